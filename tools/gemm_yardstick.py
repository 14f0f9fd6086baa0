"""Vendor yardstick for the ViT-block GEMM shapes -- COMPARISON ONLY, never on the product path ("rocBLAS/MIOpen/hipBLASLt ... may appear only
as a clearly-labelled comparison line").  torch.matmul on fp16 operands (hipBLASLt underneath) on the four ViT-block shapes at M = 8192 (two images) plus 4096³,
random uniform [-1, 1) data, interleaved rounds in one process (cdna_hip_programming.md §5.4 rules 24 / 25); next to it the library's own h8c GEMM on the same
shapes (planes in, fp32 out, no activation: the plainest epilogue, so the comparison is about the k loop) and, for each shape, 2 x t_vendor / t_h8c: an h8c product issues
two matrix units per algorithmic product (one fp16 MFMA + half a block-scaled fp8 MFMA of twice the rate), so 1.0 means "this loop is as good as the vendor's
fp16 loop", and whatever is missing from 1.0 is the loop, not the split arithmetic.
python tools/gemm_yardstick.py [--rounds 30]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--no-lib", action="store_true")
args = ap.parse_args()
dev = "cuda:0"
SHAPES = [("lin1", 8192, 4096, 1024), ("qkv", 8192, 3072, 1024), ("lin2", 8192, 1024, 4096), ("proj", 8192, 1024, 1024), ("4096^3", 4096, 4096, 4096),
          ("8192^3", 8192, 8192, 8192)]
torch.manual_seed(0)
ops = {}
for name, M, N, K in SHAPES:
    a = (torch.rand(M, K, device=dev) * 2 - 1)
    w = (torch.rand(N, K, device=dev) * 2 - 1)
    ops[name] = dict(M=M, N=N, K=K, a16=a.half(), w16=w.half(), abf=a.bfloat16(), wbf=w.bfloat16(), a=a, w=w)

lib_ok = False
if not args.no_lib:
    try:
        import mmsa  # noqa: E402,F401
        from mmsa import ops as mops  # noqa: E402
        lib_ok = True
    except Exception as e:  # the yardstick still runs without the library
        print("library not loaded:", e)
if lib_ok:
    for name, M, N, K in SHAPES:
        o = ops[name]
        o["Ap"] = mops.split_planes(o["a"], fmt=mops.FMT_H8C)
        o["Wp"] = mops.split_planes(o["w"], fmt=mops.FMT_H8C)
        o["out"] = torch.empty(M, N, device=dev)
        o["outp"] = mops.alloc_planes(M, N, dev, fmt=mops.FMT_H8C)
        o["Ab"] = mops.split_planes(o["a"], fmt=mops.FMT_F3)          # fp16 hi/lo pairs: 3 fp16 MFMAs per product
        o["Wb"] = mops.split_planes(o["w"], fmt=mops.FMT_F3, weight=True)


def time_ms(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


variants = []
for name, M, N, K in SHAPES:
    o = ops[name]
    variants.append((name, "vendor fp16", (lambda o=o: torch.matmul(o["a16"], o["w16"].t()))))
    variants.append((name, "vendor bf16", (lambda o=o: torch.matmul(o["abf"], o["wbf"].t()))))
    if lib_ok:
        variants.append((name, "h8c fp32-out", (lambda o=o: mops.gemm(o["Ap"], o["Wp"], out=o["out"]))))
        variants.append((name, "h8c planes", (lambda o=o: mops.gemm(o["Ap"], o["Wp"], out_planes=o["outp"]))))
        variants.append((name, "f3 fp32-out", (lambda o=o: mops.gemm(o["Ab"], o["Wb"], out=o["out"]))))
for _, _, fn in variants:   # warm-up + hipBLASLt heuristics
    for _ in range(3):
        fn()
torch.cuda.synchronize()
res = {}
for r in range(args.rounds):
    for name, kind, fn in variants:
        res.setdefault((name, kind), []).append(time_ms(fn, 4))
print(f"{'shape':8s} {'M':>6s} {'N':>6s} {'K':>6s}  {'kind':12s} {'median us':>10s} {'min us':>8s} {'TFLOP/s(med)':>12s} {'of 2.5 PF':>9s}")
for name, M, N, K in SHAPES:
    tv = None
    for kind in ("vendor fp16", "vendor bf16", "h8c fp32-out", "h8c planes", "f3 fp32-out"):
        if (name, kind) not in res:
            continue
        v = sorted(res[(name, kind)])
        med, mn = v[len(v) // 2] * 1e3, v[0] * 1e3
        tf = 2.0 * M * N * K / med / 1e6
        extra = ""
        if kind == "vendor fp16":
            tv = med
        elif kind.startswith("h8c"):
            extra = f"  2 x t_vendor / t = {2 * tv / med:.3f}"
        elif kind.startswith("f3"):
            extra = f"  3 x t_vendor / t = {3 * tv / med:.3f}"
        print(f"{name:8s} {M:6d} {N:6d} {K:6d}  {kind:12s} {med:10.1f} {mn:8.1f} {tf:12.1f} {tf / 2500:9.3f}{extra}")
