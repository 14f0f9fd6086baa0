"""Time the ConvNeXt 7x7 depthwise kernel alone at the four stage shapes of ViT-L 1024^2 (both streams batched; GPU box).
python tools/dwconv_bench.py [images per stream, default 2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda"
tot = 0.0
for (C, H, nblk) in ((96, 256, 3), (192, 128, 3), (384, 64, 27), (768, 32, 3)):
    x = torch.randn(2 * B * H * H, C, device=dev)
    w = torch.randn(2, 49, C, device=dev) * 0.1
    b = torch.randn(2, C, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        ops.dwconv(x, w, b, y, 2 * B, H, H, 7, imgs_per_group=B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(20):
            ops.dwconv(x, w, b, y, 2 * B, H, H, 7, imgs_per_group=B)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    gb = 2 * x.numel() * 4 / 1e9
    tot += best * nblk
    print(f"dwconv7 C={C:4d} {H}x{H} x {2 * B} images: {best:7.1f} us  {gb / (best * 1e-6) / 1e3:5.2f} TB/s (in + out)  x {nblk} blocks")
print(f"per forward: {tot / 1e3:.3f} ms")
