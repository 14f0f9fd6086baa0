"""Operand-precision study of the SAM attention blocks on the CPU oracle (test infrastructure, never on the product path):
which operands of Attention.forward (IE:465-501) tolerate rounding to fp16 / bf16, measured end to end on f1..f4 against the plain
fp32 oracle.  The numbers behind LAB_NOTES.md section 2 "Attention on single fp16 MFMAs".
    python tools/attention_precision_study.py [tiny256|vitb512 ...] [--logit-scale 1,4,16]
--logit-scale S: the q and k rows of every qkv projection (weights and biases) are multiplied by sqrt(S), i.e. every attention logit by
S -- the seeded test weights give logits of a few units, released SAM checkpoints have much peakier attention; the
line of each setting starts with the logit statistics it produces (std / max |logit| over the blocks, mean of the row maxima of P).
modes: p_bf16 / p_f16      softmax probabilities rounded before P V
       pv_f16              P and v in fp16 (one fp16 MFMA per P V product: v_fmt = 1)
       pv_f16_vlo8         P in fp16, v as fp16 hi + e5m2 lo
       qk_h8               Q K^T on h8 operands (fp16 hi hi + e5m2 cross terms)
       qk_f16              q and k in fp16 (one MFMA per Q K^T product)
       all_f16             q, k, v, P and the rel-pos tables in fp16 (every contraction one fp16 MFMA: v_fmt = 2)
       all_b3              every contraction on bf16 hi/lo operands, three products each (v_fmt = 0: the default of round 3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import ref_encoder as R
from tests.configs import CONFIGS, make_input
from tests.weights import seeded_state_dict

MODE = {"m": None}
STATS = {"on": False, "std": [], "max": [], "pmax": []}
MODES = ("p_bf16", "p_f16", "pv_f16", "pv_f16_vlo8", "qk_h8", "qk_f16", "all_f16", "all_b3")


def q8(t):
    return t.to(torch.float8_e5m2).float()


def b3(a, b):
    """a @ b on bf16 hi/lo operands: hi hi + hi lo + lo hi (the kernels' three MFMAs)"""
    ah, bh = a.bfloat16().float(), b.bfloat16().float()
    al, bl = (a - ah).bfloat16().float(), (b - bh).bfloat16().float()
    return ah @ bh + ah @ bl + al @ bh


def forward(self, x):
    m = MODE["m"]
    B, H, W, _ = x.shape
    qkv = self.qkv(x).reshape(B, H * W, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * self.num_heads, H * W, -1).unbind(0)
    rph, rpw = self.rel_pos_h, self.rel_pos_w
    if m == "qk_h8":
        qh, kh = q.half().float(), k.half().float()
        ql, kl = (q - qh) * 2048.0, (k - kh) * 2048.0
        attn = (qh @ kh.transpose(-2, -1) + (q8(qh) @ q8(kl).transpose(-2, -1) + q8(ql) @ q8(kh).transpose(-2, -1)) / 2048.0) * self.scale
    elif m == "all_b3":
        attn = b3(q, k.transpose(-2, -1)) * self.scale
    elif m in ("qk_f16", "all_f16"):
        attn = (q.half().float() @ k.half().float().transpose(-2, -1)) * self.scale
        if m == "all_f16":
            q, rph, rpw = q.half().float(), rph.half().float(), rpw.half().float()
    else:
        attn = (q * self.scale) @ k.transpose(-2, -1)
    attn = R.add_decomposed_rel_pos(attn, q, rph, rpw, (H, W), (H, W))
    if STATS["on"]:
        STATS["std"].append(attn.std().item()); STATS["max"].append(attn.abs().max().item())
        STATS["pmax"].append(attn.softmax(-1).amax(-1).mean().item())
    if m in (None, "qk_h8", "qk_f16"):
        x = attn.softmax(dim=-1) @ v
    else:   # the kernels' form: un-normalised exponentials, the row sum taken before the rounding
        p = torch.exp(attn - attn.amax(dim=-1, keepdim=True))
        l = p.sum(-1, keepdim=True)
        if m == "p_bf16":
            x = p.bfloat16().float() @ v
        elif m == "all_b3":
            x = b3(p, v)
        elif m == "p_f16":
            x = p.half().float() @ v
        elif m == "pv_f16_vlo8":
            vh = v.half().float()
            x = p.half().float() @ vh + q8(p) @ (q8((v - vh) * 2048.0) / 2048.0)
        else:
            x = p.half().float() @ v.half().float()
        x = x / l
    x = x.view(B, self.num_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return self.proj(x)


if __name__ == "__main__":
    R.Attention.forward = forward
    args = sys.argv[1:]
    scales = [1.0]
    if "--logit-scale" in args:
        i = args.index("--logit-scale")
        scales = [float(v) for v in args[i + 1].split(",")]
        del args[i:i + 2]
    for name in (args or ["tiny256"]):
        cfg = CONFIGS[name]
        for S in scales:
            torch.manual_seed(0)
            base = R.OracleEncoder(**cfg["kwargs"])
            sd = seeded_state_dict(base, seed=cfg["seed"])
            D = cfg["kwargs"]["embed_dim"]
            for k_ in sd:
                if k_.endswith("attn.qkv.weight") or k_.endswith("attn.qkv.bias"):
                    sd[k_] = sd[k_].clone()
                    sd[k_][:2 * D] *= S ** 0.5
            base.load_state_dict(sd)
            base.eval()
            x = make_input(cfg)
            with torch.no_grad():
                MODE["m"] = None
                STATS.update(on=True, std=[], max=[], pmax=[])
                ref, _ = base(x)
                STATS["on"] = False
                print(f"{name} logit-scale {S:g}: logit std {sum(STATS['std']) / len(STATS['std']):.2f} (max over blocks {max(STATS['std']):.2f}), "
                      f"max |logit| {max(STATS['max']):.1f}, mean row-max of P {sum(STATS['pmax']) / len(STATS['pmax']):.3f}", flush=True)
                for m in (MODES if S == 1.0 else ("qk_h8", "qk_f16", "pv_f16", "all_f16", "all_b3")):
                    MODE["m"] = m
                    out, _ = base(x)
                    errs = [((o - r).norm() / r.norm()).item() for o, r in zip(out, ref)]
                    mx = [((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out, ref)]
                    print(f"{name} S={S:g} {m:12s} rel_l2 " + " ".join(f"{e:.1e}" for e in errs) + "  max_rel " + " ".join(f"{e:.1e}" for e in mx), flush=True)
