"""Timing experiment: ConvNeXt pointwise pair (pw1 -> GELU -> pw2) at stage 0 / 1 shapes, whole tensor against row chunks whose
hidden tensor fits the 256 MiB Infinity Cache (the chunk's hidden rows are written by pw1 and read back by pw2 before they leave it).
python tools/chunk_mlp_exp.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
ops = mmsa.ops
dev = "cuda:0"
for (label, M, C) in (("stage0", 131072, 96), ("stage1", 32768, 192), ("stage2", 8192, 384)):
    Hd = 4 * C
    b = 2   # two streams, batched (per-stream weights)
    a = ops.split_planes(torch.randn(b * M, C, device=dev), kpad=ops.pad32(C))
    w1 = ops.split_planes(torch.randn(b * Hd, C, device=dev) / C ** 0.5); w1 = ops.Planes(w1.p, Hd, C, w1.kpad)
    w2 = ops.split_planes(torch.randn(b * C, Hd, device=dev) / Hd ** 0.5); w2 = ops.Planes(w2.p, C, Hd, w2.kpad)
    b1 = torch.randn(b * Hd, device=dev); b2 = torch.randn(b * C, device=dev); gam = torch.randn(b * C, device=dev)
    x = torch.randn(b * M, C, device=dev)
    hid = ops.alloc_planes(b * M, Hd, dev)

    def pair(rows, nchunk):
        # chunk c of stream s = rows [s*M + c*rows, +rows): batch stride stays M rows
        for c in range(nchunk):
            r0 = c * rows
            ops.gemm(a.rows(r0), w1, bias=b1, act="gelu", out_planes=hid.rows(r0), batch=b, m=rows, stride_a=M * 2 * a.kpad,
                     stride_w=Hd * 2 * w1.kpad, stride_bias=Hd, stride_cp=M * 2 * hid.kpad)
            ops.gemm(hid.rows(r0), w2, x[r0:], bias=b2, colscale=gam, resid=x[r0:], batch=b, m=rows, stride_a=M * 2 * hid.kpad,
                     stride_w=C * 2 * w2.kpad, stride_bias=C, stride_r=M * C, stride_c=M * C)
    res = []
    for nchunk in (1, 2, 4, 8):
        rows = M // nchunk
        for _ in range(3): pair(rows, nchunk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): pair(rows, nchunk)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
        res.append(f"{nchunk} chunk(s) {us:7.1f} us (hidden per chunk {b * rows * Hd * 4 / 2**20:6.0f} MiB)")
    print(f"{label} M={M} C={C}: " + " | ".join(res))
