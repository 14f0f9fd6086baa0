"""Time the windowed-attention kernel alone at the ViT-L shape (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
vf = os.environ.get("MMSA_WATTN_VF", "1") == "1"    # h8 planes throughout: every contraction on the fp16 MFMA (the production form)
H, W, heads, hd, ws = 64, 64, 16, 64, 14
D = heads * hd
dev = "cuda"
x, brow = torch.randn(B * H * W, 3 * D, device=dev), torch.randn(1, 3 * D, device=dev)
pf = ops.FMT_H8 if vf else ops.FMT_B3
qkv = ops.split_planes(x, fmt=pf)
bias = ops.split_planes(brow, kpad=3 * D, fmt=pf)
relp = ops.window_relpos_planes(torch.randn(27, hd, device=dev) * 0.3, torch.randn(27, hd, device=dev) * 0.3, ws, fmt=pf)
out = ops.alloc_planes(B * H * W, D, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
best = 1e9
for rnd in range(3):
    for _ in range(3):
        ops.window_attention(qkv, bias, relp, out, B, H, W, heads, hd, ws, hd ** -0.5)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        ops.window_attention(qkv, bias, relp, out, B, H, W, heads, hd, ws, hd ** -0.5)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / n * 1e3)
print(f"wattn B={B} vf={vf}: {best:.1f} us per launch")
