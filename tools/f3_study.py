"""Would fp16 hi + fp16 lo operands (22 significant bits, the same three MFMAs per product as bf16 hi/lo: hh + hl + lh on the fp16 MFMA) buy parity
margin where bf16 hi/lo (16 bits) is used today -- the TwinConvNeXt chain and the neck's 1 x 1 convs, whose error GFFM amplifies (LAB_NOTES.md section 2)?
CPU oracle emulation (test infrastructure, never on the product path): f1..f4 against the plain fp32 oracle.   python tools/f3_study.py [vitb512]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import ref_encoder as R  # noqa: E402
from tests.configs import CONFIGS, make_input  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402
from precision_study import make_linear  # noqa: E402


def f16_split(x):
    x = x.clamp(-65504.0, 65504.0)
    h = x.half().float()
    return h, (x - h).half().float()     # fp16 subnormals kept (the MFMA does not flush them)


def lin_f3(x, w, b):
    xh, xl = f16_split(x)
    wh, wl = f16_split(w)
    y = xh @ wh.t() + xh @ wl.t() + xl @ wh.t()
    return y if b is None else y + b


def patch(model, pred, fn):
    """nn.Linear and 1 x 1 / patchify nn.Conv2d (groups = 1: the convs the HIP path runs as GEMMs) whose name satisfies pred run through fn(x2d, w2d, bias)."""
    n = 0
    for name, m in model.named_modules():
        if not pred(name):
            continue
        if isinstance(m, nn.Linear):
            m.forward = (lambda x, m=m: fn(x, m.weight, m.bias)); n += 1
        elif isinstance(m, nn.Conv2d) and m.groups == 1 and m.kernel_size == (1, 1) and m.stride == (1, 1):
            def fwd(x, m=m):
                B, C, H, W = x.shape
                y = fn(x.permute(0, 2, 3, 1).reshape(-1, C), m.weight.reshape(m.out_channels, C), m.bias)
                return y.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
            m.forward = fwd; n += 1
    return n


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb512"
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    base = R.OracleEncoder(**cfg["kwargs"])
    sd = seeded_state_dict(base, seed=cfg["seed"])
    base.load_state_dict(sd)
    base.eval()
    x = make_input(cfg)
    with torch.no_grad():
        ref, _ = base(x)
    b3 = make_linear("split3")
    h8 = make_linear("h16x8_e5m2t")
    vit = lambda n: n.startswith("blocks.") or n.startswith("interactions.") or n.startswith("up")
    cnx = lambda n: n.startswith("spm.twin_conv.")
    neck = lambda n: n.startswith("spm.") and not cnx(n)
    for label, plan in [
        ("today: ViT / interactions h8, ConvNeXt + neck bf16 hi/lo", [(vit, h8), (cnx, b3), (neck, b3)]),
        ("ConvNeXt on fp16 hi/lo", [(vit, h8), (cnx, lin_f3), (neck, b3)]),
        ("ConvNeXt + neck on fp16 hi/lo", [(vit, h8), (cnx, lin_f3), (neck, lin_f3)]),
        ("everything that is bf16 hi/lo or h8 today on fp16 hi/lo", [(vit, lin_f3), (cnx, lin_f3), (neck, lin_f3)]),
        ("only the ViT / interaction sites reduced (h8), the rest fp32", [(vit, h8)]),
    ]:
        m = R.OracleEncoder(**cfg["kwargs"])
        m.load_state_dict(sd)
        m.eval()
        counts = [patch(m, pred, fn) for pred, fn in plan]
        with torch.no_grad():
            out, _ = m(x)
        errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
        mx = [((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out, ref)]
        print(f"{name} | {label:62s} | sites {counts} | rel_l2 " + " ".join(f"{e:.1e}" for e in errs) + " | max_rel " + " ".join(f"{e:.1e}" for e in mx), flush=True)
