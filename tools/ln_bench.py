"""LayerNorm micro-benchmark on the model's shapes: python tools/ln_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
ops = mmsa.ops
dev = "cuda:0"
for rows, C, mode in ((8192, 1024, "P"), (8192, 1024, "CP"), (43008, 1024, "P"), (16384, 384, "P"), (262144, 96, "P"), (65536, 192, "P")):
    x = torch.randn(rows, C, device=dev); w = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    outp = ops.alloc_planes(rows, C, dev)
    out = torch.empty(rows, C, device=dev) if "C" in mode else None
    f = lambda: ops.layernorm(x, w, b, 1e-6, out=out, out_planes=outp)
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 50 * 1e6
    nbytes = rows * C * 4 * (2 + ("C" in mode))
    ref = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-6)
    err = (ops.planes_to_float(outp) - ref).abs().max().item()
    print(f"LN rows {rows:6d} C {C:4d} out {mode:2s}: {us:6.1f} us  {nbytes / us / 1e6:5.2f} TB/s  max err {err:.1e}")
