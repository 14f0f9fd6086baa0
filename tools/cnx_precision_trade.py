"""VERDICT r05 item 5 ("spend the precision you are not using -- on the CPU oracle first"): the TwinConvNeXt chain runs on fp16 hi/lo pairs (f3: three MFMAs per
product) at 2e-7 ... 1.6e-6 against float64, 600 x under the gate.  Which of the stage-2 pointwise convs (27 of the 36 blocks: 1.6 + 1.9 ms of f3 GEMM per step)
could run a 2-unit format, and what does it cost at the outputs?  CPU oracle emulation (test infrastructure, never on the product path): ViT / interaction
sites on h8c (as shipped), ConvNeXt on f3 (as shipped) except the selected subset; f1..f4 against the plain fp32 oracle, ViT-B @ 512 (TwinConvNeXt does not
depend on the ViT size; tools/error_budget.py has the ViT-L float64 budget of the shipped formats).
    python tools/cnx_precision_trade.py [vitb512]
Formats of the subset:
  h8c     x = fp16 hi + e5m2 lo on BOTH operands, q(hi) truncated: hi.hi + q(hi).lo + lo.q(hi)         (2 units; the ViT blocks' format)
  a16w32  A = single fp16, W = fp16 hi/lo pair: hi_a.hi_w + hi_a.lo_w                                    (2 units; VERDICT's second proposal)
  a32w16  A = fp16 hi/lo pair, W = single fp16                                                            (2 units; the mirror image)"""
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
from precision_study import make_linear  # noqa: E402
from f3_study import f16_split, lin_f3, patch  # noqa: E402

PAT = re.compile(r"spm\.twin_conv\.stages_[xy]\.(\d)\.(\d+)\.pointwise_conv(\d)")


def sel(stage=None, pw=None):
    def pred(n):
        m = PAT.match(n)
        return bool(m) and (stage is None or int(m.group(1)) in stage) and (pw is None or int(m.group(3)) == pw)
    return pred


def lin_a16w32(x, w, b):
    xh = x.clamp(-65504.0, 65504.0).half().float()
    wh, wl = f16_split(w)
    y = xh @ wh.t() + xh @ wl.t()
    return y if b is None else y + b


def lin_a32w16(x, w, b):
    xh, xl = f16_split(x)
    wh = w.half().float()
    y = xh @ wh.t() + xl @ wh.t()
    return y if b is None else y + b


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
    h8c = make_linear("h16x8_e5m2t")
    b3 = make_linear("split3")
    vit = lambda n: n.startswith("blocks.") or n.startswith("interactions.") or n.startswith("up")   # noqa: E731
    cnx = lambda n: n.startswith("spm.twin_conv.")   # noqa: E731
    neck = lambda n: n.startswith("spm.") and not cnx(n)   # noqa: E731
    rows = [("shipped: ConvNeXt f3 everywhere", None, None)]
    for fmt_name, fn in (("h8c", h8c), ("a16w32", lin_a16w32), ("a32w16", lin_a32w16)):
        rows += [(f"stage 2 pw2 (K = 1536) on {fmt_name}", sel(stage=(2,), pw=2), fn),
                 (f"stage 2 pw1 + pw2 on {fmt_name}", sel(stage=(2,)), fn),
                 (f"stages 1-3 pw2 on {fmt_name}", sel(stage=(1, 2, 3), pw=2), fn)]
    for label, pred, fn in rows:
        m = R.OracleEncoder(**cfg["kwargs"])
        m.load_state_dict(sd)
        m.eval()
        n_sub = patch(m, pred, fn) if pred is not None else 0
        n_f3 = patch(m, (lambda n: cnx(n) and not (pred is not None and pred(n))), lin_f3)
        patch(m, vit, h8c)
        patch(m, neck, b3)
        with torch.no_grad():
            out, _ = m(x)
        errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
        mx = [((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out, ref)]
        print(f"{name} | {label:40s} | subset {n_sub:3d} f3 {n_f3:3d} | rel_l2 " + " ".join(f"{e:.1e}" for e in errs) + " | max_rel " + " ".join(f"{e:.1e}" for e in mx), flush=True)
