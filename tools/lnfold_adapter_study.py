"""Precision study of folding the ADAPTER-token LayerNorms into their consumer GEMMs, on the CPU oracle (test
infrastructure, never on the product path).  Sites: the extractor's query_norm(c) -> sampling_offsets / attention_weights, its
ffn_norm(c) -> ConvFFN.fc1, the injector's feat_norm(c) -> value_proj -- every LayerNorm whose input is the 21 n x D adapter token
matrix c.  'today' = LN, then the h8 product; 'folded' = h8 product on the RAW c against W o w, epilogue rstd * (acc - mean * colsum) + b'
(tools/lnfold_study.py has the same study for the ViT stream, where the fold was built).  f1..f4 against the plain fp32 oracle.
    python tools/lnfold_adapter_study.py [vitb512|tiny256]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import ref_encoder as R  # noqa: E402
from tests.configs import CONFIGS, make_input  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402
from lnfold_study import h8_matmul, q8  # noqa: E402

MODE = {"m": None}
STATS = []


def ln_linear(x, ln, lin):
    m = MODE["m"]
    if m is None:
        return lin(ln(x))
    if m == "today":
        return h8_matmul(ln(x), lin.weight) + lin.bias
    mean = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + ln.eps)
    wp = lin.weight * ln.weight[None, :]
    wh = wp.half().float()
    s = (wh + q8((wp - wh) * 2048.0) / 2048.0).sum(1)
    bp = lin.weight @ ln.bias + lin.bias
    STATS.append((mean.abs() * rstd).mean().item())
    return rstd * (h8_matmul(x, wp) - mean * s) + bp


def msda_forward(self, query_raw, qnorm, reference_points, feat_raw, fnorm, spatial_shapes, level_start_index, fold_q, fold_f):
    import torch.nn.functional as F
    N, Lq, _ = query_raw.shape
    _, Lin, _ = feat_raw.shape
    hd = int(self.ratio * self.d_model) // self.n_heads
    value = (ln_linear(feat_raw, fnorm, self.value_proj) if fold_f else self.value_proj(fnorm(feat_raw))).view(N, Lin, self.n_heads, hd)
    if fold_q:
        off = ln_linear(query_raw, qnorm, self.sampling_offsets)
        aw = ln_linear(query_raw, qnorm, self.attention_weights)
    else:
        q = qnorm(query_raw)
        off, aw = self.sampling_offsets(q), self.attention_weights(q)
    off = off.view(N, Lq, self.n_heads, self.n_levels, self.n_points, 2)
    aw = F.softmax(aw.view(N, Lq, self.n_heads, self.n_levels * self.n_points), -1).view(N, Lq, self.n_heads, self.n_levels, self.n_points)
    normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    return self.output_proj(R.msda_core(value, spatial_shapes, loc, aw))


def extractor_forward(self, query, ref, feat, ss, lsi, H, W):   # query = c (adapter tokens), feat = x (ViT stream)
    query = query + msda_forward(self.attn, query, self.query_norm, ref, feat, self.feat_norm, ss, lsi, True, False)
    h = ln_linear(query, self.ffn_norm, self.ffn.fc1)
    f = self.ffn
    B, N, C = h.shape
    n = N // 21
    conv = f.dwconv.dwconv
    x1 = conv(h[:, 0:16 * n].transpose(1, 2).reshape(B, C, H * 2, W * 2)).flatten(2).transpose(1, 2)
    x2 = conv(h[:, 16 * n:20 * n].transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
    x3 = conv(h[:, 20 * n:].transpose(1, 2).reshape(B, C, H // 2, W // 2)).flatten(2).transpose(1, 2)
    return query + f.fc2(torch.nn.functional.gelu(torch.cat([x1, x2, x3], dim=1)))


def injector_forward(self, query, ref, feat, ss, lsi):          # query = x, feat = c
    return query + self.gamma * msda_forward(self.attn, query, self.query_norm, ref, feat, self.feat_norm, ss, lsi, False, True)


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
        R.Extractor.forward = extractor_forward
        R.Injector.forward = injector_forward
        MODE["m"] = None
        chk, _ = base(x)
        print("patched forward == oracle:", max(((a - b).abs().max() / b.abs().max()).item() for a, b in zip(chk, ref)))
        for m in ("today", "folded"):
            MODE["m"] = m
            STATS.clear()
            out, _ = base(x)
            errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
            mx = [((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out, ref)]
            extra = f"  mean |mean| / std over the folded LN inputs {sum(STATS) / len(STATS):.3f} (max {max(STATS):.3f})" if STATS else ""
            print(f"{name} adapter sites {m:7s} rel_l2 " + " ".join(f"{e:.1e}" for e in errs) + "  max_rel " + " ".join(f"{e:.1e}" for e in mx) + extra, flush=True)
