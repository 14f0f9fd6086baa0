"""Time the windowed-attention kernel alone at the ViT-L shape (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
B, H, W, heads, hd, ws = 2, 64, 64, 16, 64, 14
D = heads * hd
dev = "cuda"
qkv = ops.split_planes(torch.randn(B * H * W, 3 * D, device=dev))
bias = ops.split_planes(torch.randn(1, 3 * D, device=dev), kpad=3 * D)
relp = ops.window_relpos_planes(torch.randn(27, hd, device=dev) * 0.3, torch.randn(27, hd, device=dev) * 0.3, ws)
out = ops.alloc_planes(B * H * W, D, dev)
for _ in range(3):
    ops.window_attention(qkv, bias, relp, out, B, H, W, heads, hd, ws, hd ** -0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 50
for _ in range(n):
    ops.window_attention(qkv, bias, relp, out, B, H, W, heads, hd, ws, hd ** -0.5)
e1.record(); torch.cuda.synchronize()
print(f"wattn {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch (debug={os.environ.get('MMSA_WATTN_DEBUG', '0')})")
