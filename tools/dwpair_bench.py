"""Time the neck's gated pair-conv stage (mmsa_dwpair_gate) at the four levels of ViT-L 1024^2, batch 2 (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
B, dev, tot = 2, "cuda", 0.0
for (C, H) in ((192, 256), (384, 128), (768, 64), (1536, 32)):
    x = torch.randn(B * H * H, 2 * C, device=dev)
    w = torch.randn(9, C, 2, 2, device=dev) * 0.2
    outp = ops.alloc_planes(B * H * H, C, dev)
    for _ in range(3):
        ops.dwpair_gate(x, w, None, B, H, H, C, out_planes=outp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(20):
            ops.dwpair_gate(x, w, None, B, H, H, C, out_planes=outp)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    gb = (x.numel() * 4 + B * H * H * C * 4) / 1e9
    tot += best
    print(f"dwpair_gate C={C:4d} {H}x{H} x {B}: {best:7.1f} us  {gb / (best * 1e-6) / 1e3:5.2f} TB/s (compulsory in + out)")
print(f"per forward: {tot / 1e3:.3f} ms")
