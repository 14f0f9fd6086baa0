"""Precision study of folding the ConvNeXt blocks' LayerNorm (TC:103-106) into pointwise_conv1 (same scheme as tools/lnfold_study.py for
the ViT blocks), on the CPU oracle with the kernels' bf16 hi/lo arithmetic (test infrastructure, never on the product path).
'today' = LN, then the bf16 hi/lo product; 'folded' = bf16 hi/lo product on the RAW depthwise-conv output against W o w,
epilogue rstd * (acc - mean * colsum) + b'.  TwinConvNeXt is the error-sensitive chain (its error is amplified ~15 x by GFFM:
LAB_NOTES.md section 2), so what matters is the twin stage outputs and f1..f4.
    python tools/lnfold_convnext_study.py [vitb512]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import ref_encoder as R  # noqa: E402
from tests.configs import CONFIGS, make_input  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402
from precision_study import bf16_split  # noqa: E402

MODE = {"m": None}
STATS = []


def b3_matmul(x, w):
    xh, xl = bf16_split(x)
    wh, wl = bf16_split(w)
    return xh @ wh.t() + xh @ wl.t() + xl @ wh.t()


def block_forward(self, x):
    sc = x
    d = self.depthwise_conv(x).permute(0, 2, 3, 1)
    m = MODE["m"]
    ln, lin = self.norm, self.pointwise_conv1
    if m is None:
        h = lin(ln(d, channel_last=True))
    elif m == "today":
        h = b3_matmul(ln(d, channel_last=True), lin.weight) + lin.bias
    else:
        mean = d.mean(-1, keepdim=True)
        var = d.var(-1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + ln.eps)
        wp = lin.weight * ln.weight[None, :]
        wh, wl = bf16_split(wp)
        s = (wh + wl).sum(1)
        bp = lin.weight @ ln.bias + lin.bias
        STATS.append((mean.abs() * rstd).mean().item())
        h = rstd * (b3_matmul(d, wp) - mean * s) + bp
    if m is None:
        y = self.pointwise_conv2(F.gelu(h))
    else:
        y = b3_matmul(F.gelu(h), self.pointwise_conv2.weight) + self.pointwise_conv2.bias
    y = y.permute(0, 3, 1, 2).mul(self.gamma.view(1, -1, 1, 1))
    return sc + y


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
        taps0 = {}
        ref, _ = base(x, taps0)
        R.ConvNeXtBlock.forward = block_forward
        for m in ("today", "folded"):
            MODE["m"] = m
            STATS.clear()
            taps = {}
            out, _ = base(x, taps)
            errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
            tw = [((taps[f"twin{i}"] - taps0[f"twin{i}"]).norm() / taps0[f"twin{i}"].norm()).item() for i in range(4) if f"twin{i}" in taps]
            extra = f"  mean |mean| / std of the LN inputs {sum(STATS) / len(STATS):.3f} (max {max(STATS):.3f})" if STATS else ""
            print(f"{name} convnext LN {m:7s} twin0..3 " + " ".join(f"{e:.1e}" for e in tw) + "  f1..f4 " + " ".join(f"{e:.1e}" for e in errs) + extra, flush=True)
