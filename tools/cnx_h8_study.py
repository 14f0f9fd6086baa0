"""Selective h8 operands inside TwinConvNeXt, on the CPU oracle (test infrastructure, never on the product path).
Every ConvNeXt pointwise conv runs the bf16 hi/lo arithmetic of today's kernels except a selected subset, which runs the h8 arithmetic
(fp16 hi + e5m2 cross terms); f1..f4 against the plain fp32 oracle, ViT-B @ 512 (TwinConvNeXt does not depend on the ViT size).
    python tools/cnx_h8_study.py [vitb512]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import ref_encoder as R  # noqa: E402
from tests.configs import CONFIGS, make_input  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402
from precision_study import patch  # noqa: E402

PAT = re.compile(r"spm\.twin_conv\.stages_[xy]\.(\d)\.(\d+)\.pointwise_conv(\d)")


def sel(stage=None, pw=None, first=None, last=None):
    def pred(n):
        m = PAT.match(n)
        if not m:
            return False
        s, b, p = int(m.group(1)), int(m.group(2)), int(m.group(3))
        return (stage is None or s in stage) and (pw is None or p == pw) and (first is None or b >= first) and (last is None or b <= last)
    return pred


VARIANTS = [
    ("none (bf16 hi/lo everywhere)", lambda n: False),
    ("stages 1-3, pw1 + pw2 (the 'cnx' site)", sel(stage=(1, 2, 3))),
    ("stage 2, pw1 + pw2", sel(stage=(2,))),
    ("stage 2, pw2 only (K = 1536)", sel(stage=(2,), pw=2)),
    ("stage 2, pw1 only (K = 384)", sel(stage=(2,), pw=1)),
    ("stage 2, last 9 blocks, pw1 + pw2", sel(stage=(2,), first=18)),
    ("stage 2, last 9 blocks, pw2 only", sel(stage=(2,), pw=2, first=18)),
    ("stage 2, first 9 blocks, pw1 + pw2", sel(stage=(2,), last=8)),
    ("stage 3, pw1 + pw2", sel(stage=(3,))),
]

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
    cnx = lambda n: n.startswith("spm.twin_conv.")
    for label, pred in VARIANTS:
        m = R.OracleEncoder(**cfg["kwargs"])
        m.load_state_dict(sd)
        m.eval()
        n8 = patch(m, pred, "h16x8_e5m2")
        n3 = patch(m, lambda n: cnx(n) and not pred(n), "split3")
        with torch.no_grad():
            out, _ = m(x)
        errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
        mx = [((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out, ref)]
        print(f"{name} | {label:42s} | h8 sites {n8:3d} b3 sites {n3:3d} | rel_l2 " + " ".join(f"{e:.1e}" for e in errs) + " | max_rel " + " ".join(f"{e:.1e}" for e in mx), flush=True)
