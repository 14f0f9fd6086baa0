"""Does the 4-wave flavour of the h8c GEMM (gemm_h8c4.hip: 128 x 128 tiles, two workgroups per CU) help the launches whose 256-row tiling leaves a FRACTIONAL last
round (M = 43008 sites -- ConvFFN fc1 at N = 256 is 336 tiles = 1.31 rounds, the value projection 2.6, the output projection 5.25)?
Same call with GEMM_FLAVOUR = 8 and 4, interleaved, operands rotated through > 256 MiB; whole chip and with the grid capped at 128 CUs (what a chain of the
two-chain step gets, at one image: M = 21504).   python tools/h8c4_tailround_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch  # noqa: E402
import mmsa  # noqa: E402

ops = mmsa.ops
DEV = "cuda:0"
CASES = [("ConvFFN fc1      ", 43008, 256, 1024, False), ("offsets / weights", 43008, 256, 1024, False), ("value proj (c)   ", 43008, 512, 1024, False),
         ("out-proj (resid) ", 43008, 1024, 512, True), ("proj (resid)     ", 8192, 1024, 1024, True)]
ROT = 3
for cap, scale in ((0, 1), (128, 2)):
    for label, M0, N, K, use_res in CASES:
        M = M0 // scale
        aps = [ops.split_planes(torch.randn(M, K, device=DEV), fmt=ops.FMT_H8C) for _ in range(ROT)]
        wp = ops.split_planes(torch.randn(N, K, device=DEV) / K ** 0.5, fmt=ops.FMT_H8C)
        outs = [torch.randn(M, N, device=DEV) for _ in range(ROT)]
        res = {}
        ops.GEMM_MAX_GRID = cap
        for rnd in range(3):
            for nw in (8, 4):
                ops.GEMM_FLAVOUR = nw
                for i in range(ROT):
                    ops.gemm(aps[i], wp, outs[i], resid=outs[i] if use_res else None)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 6
                e0.record()
                for _ in range(n):
                    for i in range(ROT):
                        ops.gemm(aps[i], wp, outs[i], resid=outs[i] if use_res else None)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(nw, []).append(e0.elapsed_time(e1) / (n * ROT) * 1e3)
        ops.GEMM_FLAVOUR, ops.GEMM_MAX_GRID = 0, 0
        t8, t4 = min(res[8]), min(res[4])
        cus = cap or 256
        print(f"CUs {cus:3d}  {label} M={M:6d} N={N:5d} K={K:5d}  tiles256 {((M + 255) // 256) * (N // 128):4d} = {((M + 255) // 256) * (N // 128) / cus:5.2f} rounds   8-wave {t8:6.1f} us   4-wave {t4:6.1f} us   x{t8 / t4:.2f}", flush=True)
