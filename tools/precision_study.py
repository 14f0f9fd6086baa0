"""Operand-precision study on the CPU oracle (test infrastructure, never on the product path).
Emulates, per nn.Linear site group, the GEMM operand formats the HIP path could use and reports the end-to-end error of f1..f4
against the plain fp32 oracle:
  split3   : bf16 hi + bf16 lo, products hh + hl + lh                       (3 bf16 MFMAs per k-step; what gemm_v2 does today)
  h16x8    : fp16 hi + 8-bit lo; hh on the fp16 MFMA, cross terms Q8(hi) x Q8(lo) on the 2x-rate fp8 MFMA (e5m2 or e4m3, lo pre-scaled by 2^11)
  h16      : fp16 hi only (one MFMA)
python tools/precision_study.py [tiny256|vitb512] """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn as nn, torch.nn.functional as F
from oracle import ref_encoder as R
from tests.configs import CONFIGS, make_input
from tests.weights import seeded_state_dict


def bf16_split(x):
    h = x.bfloat16().float(); l = (x - h).bfloat16().float(); return h, l


def q8(x, fmt, trunc=False):
    if trunc:  # top byte of the fp16 encoding = e5m2 by truncation
        u = x.half().view(torch.int16) & -256
        return u.view(torch.half).float()
    return x.to(fmt).float()


def make_linear(mode):
    def lin(x, w, b):
        if mode == "split3":
            xh, xl = bf16_split(x); wh, wl = bf16_split(w)
            y = xh @ wh.t() + xh @ wl.t() + xl @ wh.t()
        elif mode == "h16":
            y = x.half().float() @ w.half().float().t()
        else:
            fmt = torch.float8_e5m2 if "e5m2" in mode else torch.float8_e4m3fn
            tr = mode.endswith("t")
            xh = x.half().float(); xl = x - xh; wh = w.half().float(); wl = w - wh
            s = 2.0 ** 11
            y = xh @ wh.t() + (q8(xh, fmt, tr) @ q8(wl * s, fmt).t() + q8(xl * s, fmt) @ q8(wh, fmt, tr).t()) / s
        return y if b is None else y + b
    return lin


def patch(model, pred, mode):
    f = make_linear(mode)
    n = 0
    for name, m in model.named_modules():
        if isinstance(m, nn.Linear) and pred(name):
            m.forward = (lambda x, m=m: f(x, m.weight, m.bias)); n += 1
    return n


GROUPS = {
    "vit": lambda n: n.startswith("blocks."),
    "vit+inter": lambda n: n.startswith("blocks.") or n.startswith("interactions."),
    "cnx": lambda n: n.startswith("spm.twin_conv."),
    "all-linear": lambda n: True,
}

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny256"
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    base = R.OracleEncoder(**cfg["kwargs"]); sd = seeded_state_dict(base, seed=cfg["seed"]); base.load_state_dict(sd); base.eval()
    x = make_input(cfg)
    with torch.no_grad():
        ref, _ = base(x)
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(GROUPS)
    for grp, pred in GROUPS.items():
        if grp not in only:
            continue
        for mode in ("split3", "h16x8_e5m2", "h16x8_e5m2t"):
            m = R.OracleEncoder(**cfg["kwargs"]); m.load_state_dict(sd); m.eval()
            n = patch(m, pred, mode)
            with torch.no_grad():
                out, _ = m(x)
            errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
            mx = [((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out, ref)]
            print(f"{name} {grp:10s} {mode:12s} sites {n:3d}  rel_l2 " + " ".join(f"{e:.1e}" for e in errs) + "  max_rel " + " ".join(f"{e:.1e}" for e in mx), flush=True)
