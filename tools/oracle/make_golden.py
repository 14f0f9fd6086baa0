"""Generate golden vectors by running the UNMODIFIED reference (imported through
tools/oracle/ref_import.py) in the build container.  Output: tests/golden/*.npz (data only:
inputs, seeds, expected outputs).  Re-run:  python tools/oracle/make_golden.py [--big]

Weights are not stored: they come from the build's deterministic generator
`tests.weights.seeded_state_dict(model, seed)` applied to the reference model with
load_state_dict; a checksum of the generated weights is stored so drift is detected.
"""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import ref_import  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402
from tests.configs import CONFIGS, HEAD_CONFIGS, make_head_inputs, make_input, weights_checksum, probe_index  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_msda():
    fn = ref_import.ref_functions()["msda_core"]
    out = {}
    # the reference's own known-answer fixture, OPS/test.py:16-33
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    out.update(t_shapes=shapes.numpy(), t_lsi=lsi.numpy(), t_value=value.numpy(), t_loc=loc.numpy(), t_aw=aw.numpy(),
               t_out=fn(value, shapes, loc, aw).numpy(),
               t_out64=fn(value.double(), shapes, loc.double(), aw.double()).numpy())
    # injector-like (3 levels) and extractor-like (1 level) cases with out-of-range samples
    g = torch.Generator().manual_seed(11)
    for tag, shp, Lq in (("inj", [(16, 12), (8, 6), (4, 3)], 48), ("ext", [(8, 6)], 252)):
        shapes = torch.as_tensor(shp, dtype=torch.long)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        N, M, D, L, P = 2, 4, 32, len(shp), 4
        value = torch.randn(N, S, M, D, generator=g)
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15  # ~23% of taps outside [0,1]
        aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
        out.update({f"{tag}_shapes": shapes.numpy(), f"{tag}_lsi": lsi.numpy(), f"{tag}_value": value.numpy(),
                    f"{tag}_loc": loc.numpy(), f"{tag}_aw": aw.numpy(), f"{tag}_out": fn(value, shapes, loc, aw).numpy()})
    np.savez_compressed(os.path.join(OUT, "msda.npz"), **out)


def gen_bookkeeping():
    f = ref_import.ref_functions()
    out = {}
    # window partition/unpartition on an index tensor: bit-exact token bookkeeping (IE:504-551)
    for (H, W, ws) in ((14, 14, 14), (16, 16, 14), (20, 20, 14), (64, 64, 14), (32, 32, 14)):
        idx = torch.arange(1, 2 * H * W + 1, dtype=torch.float32).view(2, H, W, 1)
        win, pad_hw = f["window_partition"](idx, ws)
        back = f["window_unpartition"](win, ws, pad_hw, (H, W))
        out[f"wp_{H}_{W}_{ws}"] = win.squeeze(-1).to(torch.int64).numpy()
        out[f"wu_{H}_{W}_{ws}"] = back.squeeze(-1).to(torch.int64).numpy()
    # get_rel_pos gather table incl. linear interpolation branch (IE:554-584)
    g = torch.Generator().manual_seed(7)
    for (q, L) in ((14, 27), (64, 127), (14, 31), (20, 31), (16, 31), (32, 127)):
        table = torch.randn(L, 8, generator=g)
        out[f"rp_in_{q}_{L}"] = table.numpy()
        out[f"rp_out_{q}_{L}"] = f["get_rel_pos"](q, q, table).numpy()
    np.savez_compressed(os.path.join(OUT, "bookkeeping.npz"), **out)


def gen_model(name, full):
    cfg = CONFIGS[name]
    ref = ref_import.build_reference(**cfg["kwargs"])
    keys = list(ref.state_dict().keys())
    shapes = [list(v.shape) for v in ref.state_dict().values()]
    sd = seeded_state_dict(ref, seed=cfg["seed"])
    ref.load_state_dict(sd)
    x = make_input(cfg)
    with torch.no_grad():
        fs, none = ref(x)
    assert none is None
    out = dict(weights_checksum=np.float64(weights_checksum(sd)), x_checksum=np.float64(x.double().abs().sum().item()))
    for i, f in enumerate(fs):
        f = f.contiguous()
        out[f"f{i+1}_shape"] = np.array(f.shape)
        out[f"f{i+1}_stats"] = np.array([f.double().mean().item(), f.double().abs().mean().item(), f.abs().max().item(),
                                         f.double().pow(2).sum().sqrt().item()])
        pi = probe_index(f.numel(), 2048, seed=100 + i)
        out[f"f{i+1}_probe"] = f.flatten()[pi].numpy()
        if full:
            out[f"f{i+1}"] = f.numpy() if i > 0 or f.shape[-1] <= 56 else f[..., ::2, ::2].contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **out)
    if name == "tiny224":
        with open(os.path.join(OUT, "state_dict_keys_tiny.txt"), "w") as fh:
            for k, s in zip(keys, shapes):
                fh.write(f"{k} {s}\n")
    if name == "vitl1024":
        with open(os.path.join(OUT, "state_dict_keys_vitl.txt"), "w") as fh:
            for k, s in zip(keys, shapes):
                fh.write(f"{k} {s}\n")
    print(name, [tuple(f.shape) for f in fs], flush=True)


def gen_head(name, full):
    """Reference SegformerHead (imported unmodified; see ref_import.build_reference_head) on seeded weights + inputs."""
    cfg = HEAD_CONFIGS[name]
    ref = ref_import.build_reference_head(**cfg["kwargs"])
    sd = seeded_state_dict(ref, seed=cfg["seed"])
    ref.load_state_dict(sd)
    xs = make_head_inputs(cfg)
    with torch.no_grad():
        y = ref(xs).contiguous()
    out = dict(weights_checksum=np.float64(weights_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})),
               shape=np.array(y.shape),
               stats=np.array([y.double().mean().item(), y.double().abs().mean().item(), y.abs().max().item()]),
               probe=y.flatten()[probe_index(y.numel(), 4096, seed=200)].numpy())
    if full:
        out["logits"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    with open(os.path.join(OUT, "state_dict_keys_head.txt"), "w") as fh:
        for k, v in ref.state_dict().items():
            fh.write(f"{k} {list(v.shape)}\n")
    print(name, tuple(y.shape), flush=True)


def gen_slide():
    """The reference's own EncoderDecoder.slide_inference on a seeded frame with a fixed toy encode_decode (tests/golden/slide.npz)."""
    from tests.configs import toy_encode_decode
    out = {}
    for tag, (hw, crop, stride) in dict(a=((90, 150), (64, 64), (40, 40)), b=((64, 100), (64, 64), (48, 48)), c=((70, 70), (64, 64), (64, 64))).items():
        g = torch.Generator().manual_seed(31)
        img = torch.randn(2, 6, hw[0], hw[1], generator=g)
        fn = toy_encode_decode(5, seed=77)
        y = ref_import.reference_slide_inference(fn, img, crop, stride, 5)
        out[f"{tag}_cfg"] = np.array(list(hw) + list(crop) + list(stride))
        out[f"{tag}_out"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "slide.npz"), **out)
    print("slide", {k: v.shape for k, v in out.items() if k.endswith("_out")}, flush=True)


def gen_ckpt():
    """TwinConvNeXt.init_weights (TC:403-443) of the reference on a seeded single-stream ConvNeXt checkpoint: which twin keys
    end up loaded, and their checksums (tests/golden/convnext_ckpt.npz)."""
    import tempfile
    from tests.configs import fake_convnext_checkpoint
    cfg = CONFIGS["tiny256"]
    ref0 = ref_import.build_reference(**cfg["kwargs"])
    twin = [(k[len("spm.twin_conv."):], tuple(v.shape)) for k, v in ref0.state_dict().items() if k.startswith("spm.twin_conv.")]
    ck = fake_convnext_checkpoint(twin)
    path = os.path.join(tempfile.gettempdir(), "mmsa_fake_convnext.pth")
    torch.save({"state_dict": ck}, path)
    torch.manual_seed(0)
    ref = ref_import.build_reference(**dict(cfg["kwargs"], checkpoint=path))
    sd = ref.state_dict()
    keys = [k for k in sd if k.startswith("spm.twin_conv.")]
    # a key counts as loaded when its tensor equals the checkpoint tensor it would be fed from
    loaded = []
    for k in keys:
        sub = k[len("spm.twin_conv."):]
        first, rest = sub.split(".", 1)
        src = (first[:-2] if first.endswith(("_x", "_y")) else first) + "." + rest
        if src in ck and ck[src].shape == sd[k].shape and torch.equal(ck[src], sd[k]):
            loaded.append(k)
    np.savez_compressed(os.path.join(OUT, "convnext_ckpt.npz"), loaded=np.array(loaded), n_twin=np.int64(len(keys)),
                        checksum=np.float64(sum(sd[k].double().abs().sum().item() for k in loaded)))
    print("convnext ckpt: loaded", len(loaded), "of", len(keys), "twin keys", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also ViT-B@512 and ViT-L@1024 (minutes, GBs of RAM)")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    gen_msda()
    gen_bookkeeping()
    for n in ("tiny224", "tiny256", "tiny320"):
        gen_model(n, full=True)
    gen_slide()
    gen_ckpt()
    gen_head("head_vitl", full=False)
    gen_head("head_odd", full=True)
    gen_head("head_tiny", full=True)
    if a.big:
        gen_model("vitb512", full=False)
        gen_model("vitl1024", full=False)


if __name__ == "__main__":
    main()
