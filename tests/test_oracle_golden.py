"""CPU: the oracle (oracle/ref_encoder.py) against golden vectors produced by the imported reference
(tools/oracle/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP path
with the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_encoder as R
from tests.configs import CONFIGS, make_input, probe_index, weights_checksum
from tests.weights import seeded_state_dict
from tests.util import assert_close


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_msda_reference_known_answer(golden_dir):
    """The reference's own fixture (ops/test.py:16-50): fp64 allclose, fp32 rtol 1e-2 / atol 1e-3."""
    g = _load(golden_dir, "msda.npz")
    shapes, lsi = torch.from_numpy(g["t_shapes"]), torch.from_numpy(g["t_lsi"])
    v, loc, aw = (torch.from_numpy(g[k]) for k in ("t_value", "t_loc", "t_aw"))
    for fn in (lambda *a: R.msda_core(a[0], a[1], a[3], a[4]), R.msda_direct):
        out64 = fn(v.double(), shapes, lsi, loc.double(), aw.double())
        assert torch.allclose(out64, torch.from_numpy(g["t_out64"]))
        out32 = fn(v, shapes, lsi, loc, aw)
        assert torch.allclose(out32, torch.from_numpy(g["t_out"]), rtol=1e-2, atol=1e-3)
        assert torch.allclose(out32, torch.from_numpy(g["t_out"]), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("tag", ["inj", "ext"])
def test_msda_out_of_range_samples(golden_dir, tag):
    g = _load(golden_dir, "msda.npz")
    shapes, lsi = torch.from_numpy(g[f"{tag}_shapes"]), torch.from_numpy(g[f"{tag}_lsi"])
    v, loc, aw = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("value", "loc", "aw"))
    assert ((loc < 0) | (loc > 1)).float().mean() > 0.1  # zero-padding taps are exercised
    ref = torch.from_numpy(g[f"{tag}_out"])
    assert torch.allclose(R.msda_core(v, shapes, loc, aw), ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(R.msda_direct(v, shapes, lsi, loc, aw), ref, rtol=1e-4, atol=1e-5)


def test_window_bookkeeping_bit_exact(golden_dir):
    g = _load(golden_dir, "bookkeeping.npz")
    for (H, W, ws) in ((14, 14, 14), (16, 16, 14), (20, 20, 14), (64, 64, 14), (32, 32, 14)):
        idx = torch.arange(1, 2 * H * W + 1, dtype=torch.float32).view(2, H, W, 1)
        win, pad_hw = R.window_partition(idx, ws)
        back = R.window_unpartition(win, ws, pad_hw, (H, W))
        assert np.array_equal(win.squeeze(-1).to(torch.int64).numpy(), g[f"wp_{H}_{W}_{ws}"])
        assert np.array_equal(back.squeeze(-1).to(torch.int64).numpy(), g[f"wu_{H}_{W}_{ws}"])


def test_rel_pos_tables(golden_dir):
    g = _load(golden_dir, "bookkeeping.npz")
    for (q, L) in ((14, 27), (64, 127), (14, 31), (20, 31), (16, 31), (32, 127)):
        out = R.get_rel_pos(q, q, torch.from_numpy(g[f"rp_in_{q}_{L}"]))
        assert np.array_equal(out.numpy(), g[f"rp_out_{q}_{L}"])


def _check_model(golden_dir, name, full):
    cfg = CONFIGS[name]
    g = _load(golden_dir, f"model_{name}.npz")
    torch.manual_seed(0)
    m = R.OracleEncoder(**cfg["kwargs"])
    sd = seeded_state_dict(m, seed=cfg["seed"])
    if cfg.get("large_mag"):   # values beyond the fp16-based operand formats' range inside the ViT blocks (tests/weights.py large_magnitude)
        from tests.weights import large_magnitude
        sd = large_magnitude(sd, cfg["large_mag"])
    assert abs(weights_checksum(sd) - float(g["weights_checksum"])) <= 1e-6 * float(g["weights_checksum"])
    m.load_state_dict(sd)
    x = make_input(cfg)
    assert abs(x.double().abs().sum().item() - float(g["x_checksum"])) < 1e-6 * float(g["x_checksum"])
    fs, none = m(x)
    assert none is None
    for i, f in enumerate(fs):
        assert list(f.shape) == list(g[f"f{i+1}_shape"])
        pi = probe_index(f.numel(), 2048, seed=100 + i)
        ref = torch.from_numpy(g[f"f{i+1}_probe"])
        got = f.contiguous().flatten()[pi]
        scale = float(g[f"f{i+1}_stats"][2])  # max |f|
        assert (got - ref).abs().max().item() <= 2e-5 * scale, f"{name} f{i+1} probes"
        st = g[f"f{i+1}_stats"]
        assert abs(f.double().abs().mean().item() - st[1]) <= 1e-5 * st[1]
        assert abs(f.double().pow(2).sum().sqrt().item() - st[3]) <= 1e-5 * st[3]
        if full:
            ref_full = torch.from_numpy(g[f"f{i+1}"])
            got_full = f if ref_full.shape == f.shape else f[..., ::2, ::2]
            assert (got_full - ref_full).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("name", ["tiny224", "tiny256", "tiny320", "tiny256_plain", "tiny256_norel", "tiny256_wide"])
def test_oracle_tiny_models(golden_dir, name):
    """224: window padding 14->14 none, rel-pos interpolation (pretrained 256); 256: pad 16->28; 320: pad 20->28; 256_plain: the
    constructor switches off -- with_cffn / use_extra_extractor / add_vit_feature = False (BK:32-34, AM:485-500, BK:91-92, BK:326); 256_norel: use_rel_pos = qkv_bias = False (IE:317,320-327);
    256_wide: post-LayerNorm channels of ~1e5 and a GELU hidden unit of 6e4 inside the ViT blocks (beyond the fp16-based operand formats' range)."""
    _check_model(golden_dir, name, full=True)


def test_oracle_vitb512(golden_dir):
    _check_model(golden_dir, "vitb512", full=False)


def test_oracle_vitl1024(golden_dir):
    """BASELINE.json configs[1] shape (DeLiVER ViT-L, 1024x1024), batch 1."""
    _check_model(golden_dir, "vitl1024", full=False)


@pytest.mark.parametrize("name,keys", [("tiny224", "state_dict_keys_tiny.txt"), ("tiny256_plain", "state_dict_keys_tiny_plain.txt"),
                                       ("tiny256_norel", "state_dict_keys_tiny_norel.txt")])
def test_state_dict_keys_match_reference(golden_dir, name, keys):
    m = R.OracleEncoder(**CONFIGS[name]["kwargs"])
    want = [l.split(" ")[0] for l in open(os.path.join(golden_dir, keys))]
    assert list(m.state_dict().keys()) == want


@pytest.mark.parametrize("name", ["head_tiny", "head_odd", "head_vitl"])
def test_oracle_head_matches_reference_golden(name, golden_dir):
    """oracle/ref_head.py vs the reference's SegformerHead (goldens from tools/oracle/make_golden.py:gen_head)."""
    from oracle import ref_head as RH
    from tests.configs import HEAD_CONFIGS, make_head_inputs
    cfg = HEAD_CONFIGS[name]
    orc = RH.OracleSegformerHead(**cfg["kwargs"])
    sd = seeded_state_dict(orc, seed=cfg["seed"])
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    assert abs(weights_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point}) - float(gold["weights_checksum"])) < 1e-6 * float(gold["weights_checksum"])
    orc.load_state_dict(sd)
    y = orc(make_head_inputs(cfg))
    assert tuple(y.shape) == tuple(gold["shape"])
    if "logits" in gold:
        assert_close(y, torch.from_numpy(gold["logits"]), tol=1e-5, what=f"oracle {name}")
    probe = y.flatten()[probe_index(y.numel(), 4096, seed=200)]
    assert_close(probe, torch.from_numpy(gold["probe"]), tol=1e-5, what=f"oracle {name} probes")
    keys = [ln.split(" ")[0] for ln in open(os.path.join(golden_dir, "state_dict_keys_head.txt"))]
    if name == "head_tiny":
        assert list(orc.state_dict().keys()) == keys


def test_oracle_slide_inference_matches_reference_method(golden_dir):
    """oracle/ref_segmentor.slide_inference vs the reference's own EncoderDecoder.slide_inference (golden slide.npz)."""
    from oracle import ref_segmentor as RS
    from tests.configs import toy_encode_decode
    gold = np.load(os.path.join(golden_dir, "slide.npz"))
    for tag in "abc":
        h, w, ch, cw, sh, sw = (int(v) for v in gold[f"{tag}_cfg"])
        img = torch.randn(2, 6, h, w, generator=torch.Generator().manual_seed(31))
        y = RS.slide_inference(toy_encode_decode(5, seed=77), img, (ch, cw), (sh, sw), 5)
        assert torch.equal(y, torch.from_numpy(gold[f"{tag}_out"])) or (y - torch.from_numpy(gold[f"{tag}_out"])).abs().max() < 1e-6


def test_oracle_whole_dim_modes_match_reference_methods(golden_dir):
    """oracle/ref_segmentor.whole_inference_dim / whole_inference_dim_cut vs the reference's own EncoderDecoder methods (golden
    whole_dim.npz: 'whole_dim' at and off the input size, 'whole_dim_cut' without rescale -- the FMB configs -- and with)."""
    import numpy as np
    import torch
    from oracle import ref_segmentor as RS
    from tests.configs import toy_encode_decode
    gold = np.load(os.path.join(golden_dir, "whole_dim.npz"))
    for tag in "abcd":
        h, w, d0, d1, c0, c1, rescale = [int(v) for v in gold[f"{tag}_cfg"]]
        g = torch.Generator().manual_seed(33)
        img = torch.randn(2, 6, h, w, generator=g)
        fn = toy_encode_decode(5, seed=78)
        y = RS.whole_inference_dim(fn, img, (d0, d1), bool(rescale)) if c0 == 0 else RS.whole_inference_dim_cut(fn, img, (d0, d1), (c0, c1), bool(rescale))
        want = torch.from_numpy(gold[f"{tag}_out"])
        assert y.shape == want.shape and torch.allclose(y, want, rtol=0, atol=1e-6), tag
    assert RS.whole_inference_dim(toy_encode_decode(5, seed=78), torch.zeros(1, 6, 64, 64), (64, 64), rescale=False) is None
