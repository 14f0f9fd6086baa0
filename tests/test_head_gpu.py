"""GPU parity of the on-device Segformer head (mmsa.SegformerHead, C ABI: nchw_to_planes / gemm / head_fuse /
tokens_to_nchw) against the goldens captured from the reference's SegformerHead and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_encoder as R
from oracle import ref_head as RH
from tests.configs import HEAD_CONFIGS, make_head_inputs, probe_index
from tests.weights import seeded_state_dict
from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(name):
    import mmsa
    cfg = HEAD_CONFIGS[name]
    orc = RH.OracleSegformerHead(**cfg["kwargs"])
    sd = seeded_state_dict(orc, seed=cfg["seed"])
    orc.load_state_dict(sd)
    head = mmsa.build_head(dict(type="SegformerHead", **cfg["kwargs"]))
    head.load_state_dict(sd)
    return cfg, orc, head.to(DEV)


@pytest.mark.parametrize("name", ["head_tiny", "head_odd"])
def test_head_matches_reference_golden_and_oracle(name):
    cfg, orc, head = build(name)
    xs = make_head_inputs(cfg)
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    out = head([x.to(DEV) for x in xs])
    assert tuple(out.shape) == tuple(gold["shape"])
    assert_close(out, torch.from_numpy(gold["logits"]), what=f"{name} vs reference golden")
    assert_close(out, orc(xs), what=f"{name} vs oracle")


def test_head_other_batch_and_determinism():
    cfg, orc, head = build("head_tiny")
    xs = make_head_inputs(cfg, batch=3, seed=77)
    a = head([x.to(DEV) for x in xs]).clone()
    b = head([x.to(DEV) for x in xs]).clone()
    assert torch.equal(a, b)
    assert_close(a, orc(xs), what="head batch 3")


def test_head_vitl_probes():
    cfg, _, head = build("head_vitl")
    gold = np.load(os.path.join(GOLD, "head_vitl.npz"))
    out = head([x.to(DEV) for x in make_head_inputs(cfg)])
    assert tuple(out.shape) == tuple(gold["shape"])
    probe = out.flatten()[probe_index(out.numel(), 4096, seed=200).to(DEV)]
    assert_close(probe, torch.from_numpy(gold["probe"]), what="head_vitl probes vs reference golden")
    st = gold["stats"]
    assert abs(out.double().abs().mean().item() - st[1]) <= 1e-3 * st[1]


def test_head_rejects():
    import mmsa
    kw = HEAD_CONFIGS["head_tiny"]["kwargs"]
    with pytest.raises(NotImplementedError):
        mmsa.build_head(dict(type="SegformerHead", **dict(kw, align_corners=True)))
    head = mmsa.build_head(dict(type="SegformerHead", **kw))
    with pytest.raises(RuntimeError):
        head(make_head_inputs(HEAD_CONFIGS["head_tiny"]))  # CPU tensors: no CPU path
    with pytest.raises(NotImplementedError):
        head.train()


def test_head_takes_planes_attached_by_the_backbone_tail():
    """backbone.emit_planes: the tail writes f1..f4 also as token-major planes (attached to the returned tensors); the head then
    skips its NCHW -> planes pass.  Same logits, bit for bit, as from the bare NCHW tensors."""
    import mmsa
    from tests.configs import CONFIGS, make_input
    cfg, hcfg = CONFIGS["tiny256"], HEAD_CONFIGS["head_tiny"]
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
    h = mmsa.build_head(dict(type="SegformerHead", **hcfg["kwargs"]))
    h.load_state_dict(seeded_state_dict(h, seed=hcfg["seed"]))
    h = h.to(DEV)
    x = make_input(cfg, batch=2).to(DEV)
    plain = h(m(x)[0]).clone()
    m.emit_planes = True
    fs = m(x)[0]
    assert all(hasattr(f, "_mmsa_planes") for f in fs)
    assert torch.equal(h(fs), plain)
    assert torch.equal(h([f.clone() for f in fs]), plain)   # clones carry no planes: the transposing path again


def test_head_through_the_reference_call_signature_and_no_aliasing():
    """EncoderDecoder._decode_head_forward_test calls `decode_head.forward_test(x, img_metas, test_cfg)` (ED:129-133); results of
    successive calls are separate tensors (the reference returns fresh tensors), and maps of an EARLIER backbone call are not
    served from planes that the backbone has overwritten since."""
    import mmsa
    from tests.configs import CONFIGS, make_input
    cfg, hcfg = CONFIGS["tiny256"], HEAD_CONFIGS["head_tiny"]
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
    h = mmsa.build_head(dict(type="SegformerHead", **hcfg["kwargs"]))
    h.load_state_dict(seeded_state_dict(h, seed=hcfg["seed"]))
    xa, xb = make_input(cfg, batch=2, seed=1).to(DEV), make_input(cfg, batch=2, seed=2).to(DEV)
    la = h.forward_test(m(xa)[0], img_metas=[{}], test_cfg=dict(mode="whole")).clone()
    lb = h(m(xb)[0]).clone()
    assert not torch.equal(la, lb)
    m.emit_planes = True
    fa = m(xa)[0]
    fb = m(xb)[0]          # overwrites the planes buffers that fa's maps point to
    out_a = h(fa)
    out_b = h(fb)
    assert out_a.data_ptr() != out_b.data_ptr()
    assert torch.equal(out_a, la), "head(feats of an earlier call) must not read the later call's planes"
    assert torch.equal(out_b, lb)
    with pytest.raises(NotImplementedError):
        h.forward_train(fa, [{}], None, None)


def test_head_writes_into_a_preallocated_slice():
    """`forward(inputs, out=...)`: the logits land in the caller's tensor (mmsa.Chains hands every chain its slice of the step's logits, so that
    no framework copy kernel sits inside a captured step); same values as the fresh-tensor form, wrong shapes / strides rejected."""
    cfg, orc, head = build("head_tiny")
    del orc
    xs = [x.to(DEV) for x in make_head_inputs(cfg)]
    ref = head(xs).clone()
    big = torch.full((3 * ref.shape[0],) + tuple(ref.shape[1:]), float("nan"), device=DEV)
    sl = big[ref.shape[0]:2 * ref.shape[0]]
    ret = head(xs, out=sl)
    assert ret.data_ptr() == sl.data_ptr()
    assert torch.equal(sl, ref)
    assert torch.isnan(big[:ref.shape[0]]).all() and torch.isnan(big[2 * ref.shape[0]:]).all()
    with pytest.raises(RuntimeError):
        head(xs, out=torch.empty(ref.shape[0], ref.shape[1], ref.shape[2] + 1, ref.shape[3], device=DEV))
    with pytest.raises(RuntimeError):
        head(xs, out=torch.empty_like(ref, dtype=torch.float64))
    with pytest.raises(RuntimeError):
        head(xs, out=torch.empty(ref.shape[0], ref.shape[1], ref.shape[2], 2 * ref.shape[3], device=DEV)[..., ::2])   # not contiguous
