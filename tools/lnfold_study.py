"""Precision study of folding the ViT blocks' LayerNorms into their consumer GEMMs, on the CPU oracle (test
infrastructure, never on the product path).

Today:   n = LN(x; w, b) -> planes -> qkv / lin1 GEMM on h8 operands.
Folded:  the residual stream x itself travels as planes (written by the proj / lin2 epilogue together with per-row sums); the consumer
         runs on W' = W o w (column scale folded into the packed weight) and its epilogue applies
              y = rstd_r * (x . W'^T - mean_r * s) + b',     s_n = sum_k W'_nk,   b' = W b + bias
         -- algebraically LN(x) W^T + bias; numerically the rounding errors of the products now scale with |x_k| instead of
         |x_k - mean|, and the subtraction cancels whatever mean_r * s_n contributes.
Reported: f1..f4 against the plain fp32 oracle for (a) today's form and (b) the folded form, both with the kernels' h8 arithmetic
(fp16 hi, e5m2 cross terms), plus the residual stream's |mean| / std statistics that decide how much cancels; optionally with a
constant added to every token (`--offset c`: a stream whose mean is c standard deviations) as the stress case.
    python tools/lnfold_study.py [vitb512|tiny256] [--offset 0,2,8]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import ref_encoder as R  # noqa: E402
from tests.configs import CONFIGS, make_input  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402

MODE = {"m": None, "offset": 0.0}
STATS = []


def q8(t):
    return t.to(torch.float8_e5m2).float()


def h8_matmul(x, w):
    """x @ w^T with the GEMM kernel's h8 arithmetic: fp16 hi . hi + e5m2 cross terms (lo pre-scaled by 2^11)."""
    x = x.clamp(-57344.0, 57344.0)
    xh, wh = x.half().float(), w.half().float()
    xl, wl = (x - xh) * 2048.0, (w - wh) * 2048.0
    return xh @ wh.t() + (q8(xh) @ q8(wl).t() + q8(xl) @ q8(wh).t()) / 2048.0


def ln_linear(x, ln, lin):
    m = MODE["m"]
    if m is None:
        return lin(ln(x))
    if m == "today":
        return h8_matmul(ln(x), lin.weight) + lin.bias
    # folded
    mean = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + ln.eps)
    wp = lin.weight * ln.weight[None, :]
    # s from the weights AS THE KERNEL SEES THEM (hi + the e5m2 image of lo): sum_k of what each x_k is multiplied with
    wh = wp.half().float()
    s = (wh + q8((wp - wh) * 2048.0) / 2048.0).sum(1)
    bp = lin.weight @ ln.bias + lin.bias
    acc = h8_matmul(x, wp)
    STATS.append((mean.abs() * rstd).mean().item())
    return rstd * (acc - mean * s) + bp


def block_forward(self, x, H, W):
    x = x + MODE["offset"] * 0.0   # (the offset is injected once, at the stream's entry: see main)
    x = x.unflatten(1, (H, W))
    shortcut = x
    a = self.attn
    B = x.shape[0]
    # norm1 -> qkv (window padding happens between them in the reference: pad tokens are zeros AFTER the norm, i.e. qkv = bias)
    xq = ln_linear(x, self.norm1, a.qkv)
    if self.window_size > 0:
        ws = self.window_size
        ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
        if ph or pw:
            full = a.qkv.bias.expand(B, H + ph, W + pw, -1).clone()
            full[:, :H, :W] = xq
            xq = full
        Hp, Wp = H + ph, W + pw
        xq = xq.view(B, Hp // ws, ws, Wp // ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, xq.shape[-1])
        hh, ww = ws, ws
    else:
        hh, ww = H, W
    Bw = xq.shape[0]
    qkv = xq.reshape(Bw, hh * ww, 3, a.num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, Bw * a.num_heads, hh * ww, -1).unbind(0)
    attn = (q * a.scale) @ k.transpose(-2, -1)
    attn = R.add_decomposed_rel_pos(attn, q, a.rel_pos_h, a.rel_pos_w, (hh, ww), (hh, ww))
    o = (attn.softmax(-1) @ v).view(Bw, a.num_heads, hh, ww, -1).permute(0, 2, 3, 1, 4).reshape(Bw, hh, ww, -1)
    o = a.proj(o)
    if self.window_size > 0:
        o = R.window_unpartition(o, self.window_size, (Hp, Wp), (H, W))
    x = shortcut + o
    h = F.gelu(ln_linear(x, self.norm2, self.mlp.lin1))
    x = x + self.mlp.lin2(h)
    return x.flatten(1, 2)


if __name__ == "__main__":
    args = sys.argv[1:]
    offsets = [0.0]
    if "--offset" in args:
        i = args.index("--offset")
        offsets = [float(v) for v in args[i + 1].split(",")]
        del args[i:i + 2]
    name = args[0] if args else "vitb512"
    cfg = CONFIGS[name]
    R.Block.forward = block_forward
    for off in offsets:
        torch.manual_seed(0)
        base = R.OracleEncoder(**cfg["kwargs"])
        sd = seeded_state_dict(base, seed=cfg["seed"])
        # a stream whose every token carries a common offset of `off` (in units of the pos-embed's own scale, ~ the stream's std)
        sd["pos_embed"] = sd["pos_embed"] + off * sd["pos_embed"].std()
        base.load_state_dict(sd)
        base.eval()
        x = make_input(cfg)
        with torch.no_grad():
            MODE["m"] = None
            ref, _ = base(x)
            for m in ("today", "folded"):
                MODE["m"] = m
                STATS.clear()
                out, _ = base(x)
                errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
                mx = [((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out, ref)]
                extra = f"  mean|mean|/std over the LN inputs {sum(STATS) / len(STATS):.3f} (max {max(STATS):.3f})" if STATS else ""
                print(f"{name} offset {off:g} {m:7s} rel_l2 " + " ".join(f"{e:.1e}" for e in errs) + "  max_rel " + " ".join(f"{e:.1e}" for e in mx) + extra, flush=True)
