"""Timing experiment: does a row stride that is NOT a multiple of 4 KiB (one extra 128-byte line per row) speed up the split3 GEMM?
Power-of-two row strides put every row of a k-tile on the same L2 channel.  Operand contents are random bits (timing only).
MMSA_GEMM_WPAD=<bf16 elements> pads the weight rows inside the kernel; A is padded through its view.
python tools/gemm_pad_exp.py <apad elements>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
ops = mmsa.ops
apad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wpad = int(os.environ.get("MMSA_GEMM_WPAD", "0"))
SHAPES = [("lin1", 8192, 4096, 1024, "gelu", "P", 0), ("lin2", 8192, 1024, 4096, "none", "C", 1),
          ("qkv", 8192, 3072, 1024, "none", "P", 0), ("proj", 8192, 1024, 1024, "none", "C", 1),
          ("ext out", 43008, 1024, 512, "none", "C", 1), ("up", 32768, 4096, 1024, "none", "C", 0)]
dev = "cuda:0"
res = []
for (label, M, N, K, act, outk, resid) in SHAPES:
    ab = (torch.randn(M, K + apad // 2, device=dev) * 0.5).bfloat16().view(torch.int16).repeat(1, 2)   # finite bf16 bit patterns
    a = ops.Planes(ab[:, :2 * K], M, K, K)
    wb = (torch.randn(N, K + wpad // 2, device=dev) * 0.02).bfloat16().view(torch.int16).repeat(1, 2)
    w = ops.Planes(wb, N, K, K)
    bias = torch.randn(N, device=dev)
    kw = {}
    if outk == "P":
        kw.update(out_planes=ops.alloc_planes(M, N, dev))
    else:
        c = torch.randn(M, N, device=dev)
        kw.update(out=c)
        if resid:
            kw.update(resid=c)
    for _ in range(3):
        ops.gemm(a, w, bias=bias, act=act, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        ops.gemm(a, w, bias=bias, act=act, **kw)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 30 * 1e6
    res.append(f"{label} {us:7.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF")
print(f"apad={apad} wpad={wpad} dbg={os.environ.get('MMSA_GEMM_DEBUG', '0')}: " + " | ".join(res))
