"""Time the 3 x 3 depthwise conv at its shapes in ViT-L 1024^2, batch 2: the ConvFFN's (C = 256 on the three token levels, GELU, h8 planes out) and the
neck MobileNetV2's (C = 192 .. 1536, ReLU6, bf16 hi/lo planes out).  With a debug-knob build of conv.hip in MMSA_LIB, MMSA_DWCONV3_STRIP=0 selects the
one-pixel kernel.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
B, dev, tot = 2, "cuda", 0.0
shapes = [("convffn", 256, 128, "gelu", ops.FMT_H8, 6), ("convffn", 256, 64, "gelu", ops.FMT_H8, 6), ("convffn", 256, 32, "gelu", ops.FMT_H8, 6),
          ("neck", 192, 256, "relu6", ops.FMT_B3, 5), ("neck", 384, 128, "relu6", ops.FMT_B3, 5), ("neck", 768, 64, "relu6", ops.FMT_B3, 5), ("neck", 1536, 32, "relu6", ops.FMT_B3, 5)]
for (tag, C, H, act, fmt, n_fwd) in shapes:
    x = torch.randn(B * H * H, C, device=dev)
    w = torch.randn(9, C, device=dev) * 0.2
    b = torch.randn(C, device=dev)
    outp = ops.alloc_planes(B * H * H, C, dev, fmt=fmt)
    for _ in range(3):
        ops.dwconv(x, w, b, None, B, H, H, 3, act=act, out_planes=outp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(20):
            ops.dwconv(x, w, b, None, B, H, H, 3, act=act, out_planes=outp)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    gb = 2 * x.numel() * 4 / 1e9
    tot += best * n_fwd
    print(f"dwconv3 {tag:8s} C={C:4d} {H}x{H} x {B}: {best:7.1f} us  {gb / (best * 1e-6) / 1e3:5.2f} TB/s (in + out)  x {n_fwd} per forward")
print(f"per forward: {tot / 1e3:.3f} ms")
