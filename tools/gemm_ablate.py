"""Per-shape ablation of the split3 GEMM on the model's own launch configurations (batch, activation, output kind, residual):
python tools/gemm_ablate.py            -> one table per MMSA_GEMM_DEBUG mode (each mode in its own process)
modes: 0 = full kernel, 10 = epilogue without its global stores, 1 = no global stores + no epilogue arithmetic, 2 = no epilogue.
The release library has no such knob: the workers load ab/libmmsa_knobs.so, a debug-knob build of gemm_v2.hip (tools/build_variant.sh
... -DMMSA_DEBUG_KNOBS, built here on first use; MMSA_LIB points mmsa.lib at it)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))

SHAPES = [  # (label, M, N, K, batch, act, out, resid)
    ("lin1", 8192, 4096, 1024, 1, "gelu", "P", 0), ("lin2", 8192, 1024, 4096, 1, "none", "C", 1),
    ("qkv", 8192, 3072, 1024, 1, "none", "P", 0), ("proj", 8192, 1024, 1024, 1, "none", "C", 1),
    ("cnx2 pw1", 8192, 1536, 384, 2, "gelu", "P", 0), ("cnx2 pw2", 8192, 384, 1536, 2, "none", "C", 1),
    ("ext out", 43008, 1024, 512, 1, "none", "C", 1), ("ffn fc2", 43008, 1024, 256, 1, "none", "C", 1),
    ("ffn fc1", 43008, 256, 1024, 1, "none", "C", 0), ("msda oa", 43008, 192, 1024, 1, "none", "C", 0),
    ("cnx0 pw1", 131072, 384, 96, 2, "gelu", "P", 0), ("cnx0 pw2", 131072, 96, 384, 2, "none", "C", 1),
    ("cnx1 pw1", 32768, 768, 192, 2, "gelu", "P", 0), ("cnx1 pw2", 32768, 192, 768, 2, "none", "C", 1),
    ("cnx3 pw1", 2048, 3072, 768, 2, "gelu", "P", 0), ("cnx3 pw2", 2048, 768, 3072, 2, "none", "C", 1),
]


def worker():
    import torch
    import mmsa
    ops = mmsa.ops
    dev = "cuda:0"
    res = []
    for (label, M, N, K, b, act, outk, resid) in SHAPES:
        want = os.environ.get("MMSA_ABLATE_FMT", "b3")   # operand format: b3 | h8 | h8c (random normal data either way: the clock the chip holds depends on the data)
        h8 = want == "h8" and K % 64 == 0
        fmt = ops.FMT_H8C if (want == "h8c" and K % 64 == 0) else ops.FMT_H8 if h8 else ops.FMT_B3
        a = ops.split_planes(torch.randn(b * M, K, device=dev), kpad=K, fmt=fmt)
        w = ops.split_planes(torch.randn(b * N, K, device=dev) / K ** 0.5, fmt=fmt, weight=h8)
        w = ops.Planes(w.p, N, K, w.kpad, fmt, h8)   # the first batch's view (batch stride passed explicitly)
        bias = torch.randn(b * N, device=dev)
        kw = {}
        if outk == "P":
            op = ops.alloc_planes(b * M, N, dev, fmt=fmt)
            kw.update(out_planes=op, stride_cp=op.batch_stride(M))
        else:
            c = torch.randn(b * M, N, device=dev)
            kw.update(out=c, stride_c=M * N)
            if resid:
                kw.update(resid=c, stride_r=M * N)
        kw.update(batch=b, m=M, stride_a=a.batch_stride(M), stride_w=w.batch_stride(N), stride_bias=N)
        for _ in range(3):
            ops.gemm(a, w, bias=bias, act=act, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 30
        for _ in range(reps):
            ops.gemm(a, w, bias=bias, act=act, **kw)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        res.append(us)
    print(" ".join(f"{u:8.1f}" for u in res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
        sys.exit(0)
    print("mode      " + " ".join(f"{s[0]:>8s}" for s in SHAPES))
    flops = [2.0 * s[1] * s[2] * s[3] * s[4] for s in SHAPES]
    knobs = os.path.join(ROOT, "ab", os.environ.get("MMSA_ABLATE_LIB", "libmmsa_knobs.so"))
    csrc = os.path.join(ROOT, "multimodal-sam-adapter_amd", "csrc")
    if not os.path.exists(knobs) or os.path.getmtime(knobs) < max(os.path.getmtime(os.path.join(csrc, f)) for f in ("gemm_v2.hip", "gemm_h8c.hip", "gemm_v2_epilogue.inc")):
        subprocess.run(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "ab/" + os.path.basename(knobs), "gemm_v2.hip,gemm_h8c.hip", "-DMMSA_DEBUG_KNOBS"]
                       + os.environ.get("MMSA_ABLATE_DEFS", "").split(), check=True)
    for mode in (sys.argv[1:] or ["0", "10", "1", "2"]):
        env = dict(os.environ, MMSA_GEMM_DEBUG=mode, MMSA_LIB=knobs)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
        print(f"dbg {mode:>2s} us " + line)
        if mode == "0":
            try:
                print("      TF/s " + " ".join(f"{f / float(u) / 1e6:8.1f}" for f, u in zip(flops, line.split())))
            except ValueError:
                pass
