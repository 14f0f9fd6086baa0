"""ConvNeXt pointwise pair at the stage 0 / 1 shapes: the fused kernel (csrc/mlp_fused.hip) against the two GEMM launches.
python tools/mlp_fused_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
ops = mmsa.ops
dev = "cuda:0"
for (label, M, C) in (("stage0 B=2", 131072, 96), ("stage0 B=1", 65536, 96)):
    Hd, b = 4 * C, 2
    a = ops.split_planes(torch.randn(b * M, C, device=dev), kpad=ops.pad32(C))
    w1 = ops.split_planes(torch.randn(b * Hd, C, device=dev) / C ** 0.5); w1 = ops.Planes(w1.p, Hd, C, w1.kpad)
    w2 = ops.split_planes(torch.randn(b * C, Hd, device=dev) / Hd ** 0.5); w2 = ops.Planes(w2.p, C, Hd, w2.kpad)
    b1 = torch.randn(b * Hd, device=dev); b2 = torch.randn(b * C, device=dev); gam = torch.randn(b * C, device=dev) * 0.1
    x = torch.randn(b * M, C, device=dev)
    hid = ops.alloc_planes(b * M, Hd, dev)

    def pair():
        ops.gemm(a, w1, bias=b1, act="gelu", out_planes=hid, batch=b, m=M, stride_a=M * 2 * a.kpad, stride_w=Hd * 2 * w1.kpad, stride_bias=Hd, stride_cp=M * 2 * hid.kpad)
        ops.gemm(hid, w2, x, bias=b2, colscale=gam, resid=x, batch=b, m=M, stride_a=M * 2 * hid.kpad, stride_w=C * 2 * w2.kpad, stride_bias=C, stride_r=M * C, stride_c=M * C)

    def fused():
        ops.convnext_mlp_fused(a, w1, w2, b1, b2, gam, x, M, batch=b, stride_a=M * 2 * a.kpad, stride_w1=Hd * 2 * w1.kpad, stride_w2=C * 2 * w2.kpad, stride_x=M * C)
    res = []
    for name, f in (("pair", pair), ("fused", fused)):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); res.append(f"{name} {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us")
    print(f"{label} M={M} C={C}: " + " | ".join(res), flush=True)
