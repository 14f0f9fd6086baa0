"""Micro-benchmark of the split3 GEMM kernel on the ViT-L / ConvNeXt shapes of the encoder (GPU box only).
Prints algorithmic TFLOP/s (2*M*N*K / time); the kernel issues 3x that on the MFMA pipe."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch  # noqa: E402
import mmsa  # noqa: E402

ops = mmsa.ops
DEV = "cuda:0"
SHAPES = [  # (M, N, K, planes_in, label)
    (8192, 3072, 1024, True, "qkv"), (8192, 1024, 1024, True, "proj"), (8192, 4096, 1024, True, "lin1"),
    (8192, 1024, 4096, True, "lin2"), (131072, 384, 96, True, "cnx pw1 s0"), (131072, 96, 384, True, "cnx pw2 s0"),
    (8192, 1536, 384, True, "cnx pw1 s2"), (43008, 512, 1024, True, "inj value"), (43008, 1024, 512, True, "ext out"),
    (32768, 4096, 1024, True, "up"), (131072, 1024, 192, False, "fc1 fp32-A"),
]


def bench(M, N, K, planes, reps=20):
    a = torch.randn(M, K, device=DEV)
    w = ops.split_planes(torch.randn(N, K, device=DEV) / K ** 0.5)
    b = torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    ain = ops.split_planes(a, kpad=K) if planes else a
    kw = dict(out=out)
    if os.environ.get("GEMM_BENCH_OUT") == "planes":   # what qkv / lin1 / pw1 do in the model: planes-only output
        kw = dict(out_planes=ops.alloc_planes(M, N, DEV))
    for _ in range(3):
        ops.gemm(ain, w, bias=b, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.gemm(ain, w, bias=b, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return 2.0 * M * N * K / dt / 1e12, dt * 1e6


if __name__ == "__main__":
    tot_f, tot_t = 0.0, 0.0
    for (M, N, K, pl, label) in SHAPES:
        tf, us = bench(M, N, K, pl)
        tot_f += 2.0 * M * N * K
        tot_t += us
        print(f"{label:14s} M={M:6d} N={N:5d} K={K:5d} planes={int(pl)}  {us:9.1f} us  {tf:7.1f} TFLOP/s (x3 = {3*tf:6.0f} MFMA)")
    print(f"aggregate {tot_f / tot_t / 1e6:.1f} TFLOP/s")
