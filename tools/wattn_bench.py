"""Time the windowed-attention kernel alone at the ViT-L shape (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
vf = os.environ.get("MMSA_WATTN_VF", "1") == "1"    # v columns as h8 planes: P V on the fp16 MFMA (the production form)
H, W, heads, hd, ws = 64, 64, 16, 64, 14
D = heads * hd
dev = "cuda"
x, brow = torch.randn(B * H * W, 3 * D, device=dev), torch.randn(1, 3 * D, device=dev)
qkv = ops.split_planes_qkv(x, D) if vf else ops.split_planes(x)
bias = ops.split_planes_qkv(brow, D) if vf else ops.split_planes(brow, kpad=3 * D)
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
print(f"wattn B={B} vf={vf}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch (debug={os.environ.get('MMSA_WATTN_DEBUG', '0')})")
