"""Standalone timing of the neck's skinny GEMM launches (M = 131072 at two images, N and K <= 192): the streaming kernel (gemm_stream.hip: taken by shape) against the
tiled LDS-DMA kernel (`GEMM_FLAVOUR = 4` keeps a launch on its 4-wave flavour, the one these shapes took), same call, interleaved, operands rotated through > 256 MiB so that nothing is served from the
Infinity Cache.  Prints microseconds and compulsory bytes / time.   python tools/gemm_stream_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch  # noqa: E402
import mmsa  # noqa: E402

ops = mmsa.ops
DEV = "cuda:0"
CASES = [  # (label, M, N, K, act, resid, out)
    ("MobileNetV2 2c->c  (P, resid)", 131072, 96, 192, "none", True, "P"),
    ("MobileNetV2 c->2c  (C, relu6)", 131072, 192, 96, "relu6", False, "C"),
]
ROT = 4
for label, M, N, K, act, use_res, outk in CASES:
    aps = [ops.split_planes(torch.randn(M, K, device=DEV)) for _ in range(ROT)]
    wp = ops.split_planes(torch.randn(N, K, device=DEV) / K ** 0.5)
    ress = [torch.randn(M, N, device=DEV) for _ in range(ROT)] if use_res else [None] * ROT
    outs = [torch.empty(M, N, device=DEV) for _ in range(ROT)] if outk == "C" else [None] * ROT
    outps = [ops.alloc_planes(M, N, DEV) for _ in range(ROT)] if outk == "P" else [None] * ROT
    by = 4.0 * (M * K + N * K) + 4.0 * M * N * (1 + (1 if use_res else 0))
    res = {}
    for rnd in range(3):
        for nw in (0, 4):   # 4: the tiled flavour these launches took before (K <= 256: 128-row tiles, two workgroups per CU)
            ops.GEMM_FLAVOUR = nw
            for i in range(ROT):
                ops.gemm(aps[i], wp, outs[i], act=act, resid=ress[i], out_planes=outps[i], alpha=0.5)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                for i in range(ROT):
                    ops.gemm(aps[i], wp, outs[i], act=act, resid=ress[i], out_planes=outps[i], alpha=0.5)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (n * ROT) * 1e3
            res.setdefault(nw, []).append(us)
    ops.GEMM_FLAVOUR = 0
    s_, t_ = min(res[0]), min(res[4])
    print(f"{label}  M={M} N={N:3d} K={K:3d}   stream {s_:6.1f} us = {by / s_ / 1e6:5.2f} TB/s ({by / s_ / 1e6 / 8:.3f} of 8)   tiled {t_:6.1f} us = {by / t_ / 1e6:5.2f} TB/s   x{t_ / s_:.2f}", flush=True)
