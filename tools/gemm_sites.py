"""Stand-alone timing of the model's GEMM SITES with their real epilogue configuration (operand format, LayerNorm-fold extras, plane
format of the output, residual, per-column scale, batch) -- what tools/gemm_shapes.py measures inside a forward, without the forward:
    python tools/gemm_sites.py [--rounds 3] [--only lin1,qkv] [lib.so[:VAR=val+VAR2=val] ...]        (no library: the in-tree one)
Each library is timed in its own process (MMSA_LIB), the libraries interleaved over --rounds rounds; random-normal operands (the clock the
chip holds depends on the data).  us per launch, median of the rounds; ViT-L 1024^2, batch 2 (BASELINE configs[1]).
Reference call sites: IE:154-167,488,499 (lin1 / lin2 / qkv / proj), TC:107-111 (pw1 / pw2), AM:447-451 (fc1 / fc2),
ops/modules/ms_deform_attn.py:103-129 (value / offsets / output projections)."""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))

# (label, M, N, K, batch, operand fmt, act, out kind, plane fmt out, resid, strip sums, row-norm, colscale)
SITES = [
    ("lin1", 8192, 4096, 1024, 1, "h8c", "gelu", "P", "h8c", 0, 0, 1, 0),
    ("qkv", 8192, 3072, 1024, 1, "h8c", "none", "P", "h8", 0, 0, 1, 0),
    ("lin2", 8192, 1024, 4096, 1, "h8c", "none", "CP", "h8c", 1, 1, 0, 0),
    ("proj", 8192, 1024, 1024, 1, "h8c", "none", "CP", "h8c", 1, 1, 0, 0),
    ("cnx2pw1", 8192, 1536, 384, 2, "f3", "gelu", "P", "f3", 0, 0, 0, 0),
    ("cnx2pw2", 8192, 384, 1536, 2, "f3", "none", "C", None, 1, 0, 0, 1),
    ("cnx1pw1", 32768, 768, 192, 2, "f3", "gelu", "P", "f3", 0, 0, 0, 0),
    ("cnx1pw2", 32768, 192, 768, 2, "f3", "none", "C", None, 1, 0, 0, 1),
    ("cnx3pw1", 2048, 3072, 768, 2, "f3", "gelu", "P", "f3", 0, 0, 0, 0),
    ("cnx3pw2", 2048, 768, 3072, 2, "f3", "none", "C", None, 1, 0, 0, 1),
    ("extout", 43008, 1024, 512, 1, "h8c", "none", "C", None, 1, 0, 0, 0),
    ("ffnfc2", 43008, 1024, 256, 1, "h8", "none", "C", None, 1, 0, 0, 0),
    ("ffnfc1", 43008, 256, 1024, 1, "h8c", "none", "C", None, 0, 0, 0, 0),
    ("injval", 43008, 512, 1024, 1, "h8c", "none", "C", None, 0, 0, 0, 0),
    ("msdaoa", 43008, 192, 1024, 1, "h8c", "none", "C", None, 0, 0, 0, 0),
    ("injoa", 8192, 576, 1024, 1, "h8c", "none", "C", None, 0, 0, 0, 0),
    ("extval", 8192, 512, 1024, 1, "h8c", "none", "C", None, 0, 0, 0, 0),
    ("injout", 8192, 1024, 512, 1, "h8c", "none", "CP", "h8c", 1, 1, 0, 1),
    # diagnostic variants of lin1 (not model sites; run with --only): what GELU and the row-normalising form cost its epilogue
    ("lin1none", 8192, 4096, 1024, 1, "h8c", "none", "P", "h8c", 0, 0, 1, 0),
    ("lin1norn", 8192, 4096, 1024, 1, "h8c", "gelu", "P", "h8c", 0, 0, 0, 0),
    ("lin1bare", 8192, 4096, 1024, 1, "h8c", "none", "P", "h8c", 0, 0, 0, 0),
]


def worker(only):
    import torch
    import mmsa
    ops = mmsa.ops
    dev = "cuda:0"
    FM = {"b3": ops.FMT_B3, "h8": ops.FMT_H8, "h8c": ops.FMT_H8C, "f3": ops.FMT_F3}
    res = []
    for (label, M, N, K, b, fmt_n, act, outk, pf, resid, rs, rn, cs) in SITES:
        if (only and label not in only) or (not only and label in ("lin1none", "lin1norn", "lin1bare")):
            res.append(float("nan"))
            continue
        fmt = FM[fmt_n]
        h8w = fmt == ops.FMT_H8
        a = ops.split_planes(torch.randn(b * M, K, device=dev), kpad=K, fmt=fmt)
        wa = ops.split_planes(torch.randn(b * N, K, device=dev) / K ** 0.5, fmt=fmt, weight=h8w)
        w = ops.Planes(wa.p, N, K, wa.kpad, fmt, h8w)
        kw = dict(bias=torch.randn(b * N, device=dev), act=act, batch=b, m=M, stride_a=a.batch_stride(M), stride_w=wa.batch_stride(N), stride_bias=N)
        if cs:
            kw.update(colscale=torch.rand(b * N, device=dev) + 0.5)
        if "C" in outk:
            c = torch.randn(b * M, N, device=dev)
            kw.update(out=c, stride_c=M * N)
            if resid:
                kw.update(resid=c, stride_r=M * N)
        if "P" in outk:
            op = ops.alloc_planes(b * M, N, dev, fmt=FM[pf])
            kw.update(out_planes=op, stride_cp=op.batch_stride(M))
        if rs:
            kw.update(rowstats_out=torch.empty(b * M, 2 * (N // 64), device=dev))
        if rn:
            mr = torch.stack([torch.randn(b * M, device=dev) * 0.1, torch.rand(b * M, device=dev) + 0.5], 1).contiguous()
            kw.update(row_norm=(mr, torch.randn(b * N, device=dev)))
        for _ in range(3):
            ops.gemm(a, w, **kw)
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.gemm(a, w, **kw)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e6)
        del a, wa, w, kw
    print("RESULT " + " ".join(f"{u:.2f}" for u in res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(set(sys.argv[2:]))
        sys.exit(0)
    args = sys.argv[1:]
    rounds, only = 3, []
    while args and args[0].startswith("--"):
        if args[0] == "--rounds":
            rounds = int(args[1]); args = args[2:]
        elif args[0] == "--only":
            only = args[1].split(","); args = args[2:]
        else:
            raise SystemExit(f"unknown flag {args[0]}")
    libs = args or [""]
    runs = {lib: [] for lib in libs}
    for rnd in range(rounds):
        for lib in libs:
            env = dict(os.environ)
            path, _, sets = lib.partition(":")      # "ab/lib.so:VAR=1+VAR2=x": environment of that library's worker (debug-knob builds)
            env.update(dict(kv.split("=", 1) for kv in sets.split("+") if kv))
            if path:
                env["MMSA_LIB"] = os.path.join(ROOT, path) if not os.path.isabs(path) else path
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"] + only, env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(f"{lib or 'in-tree'}: FAILED\n{out.stderr[-1500:]}", flush=True)
                continue
            runs[lib].append([float(x) for x in line[0].split()[1:]])
    print(f"{'site':9s} {'M x N x K (batch)':26s} " + " ".join(f"{(os.path.basename(l.replace('libmmsa_', '').replace('.so', '')) or 'in-tree')[:18]:>18s}" for l in libs) + ("   ratio to first" if len(libs) > 1 else ""))
    tot = {lib: 0.0 for lib in libs}
    for i, s in enumerate(SITES):
        med = {lib: (statistics.median(r[i] for r in runs[lib]) if runs[lib] else float("nan")) for lib in libs}
        if med[libs[0]] != med[libs[0]]:
            continue
        flops = 2.0 * s[1] * s[2] * s[3] * s[4]
        cells = " ".join(f"{med[l]:9.1f} {flops / med[l] / 1e6:6.0f}TF" for l in libs)
        ratio = "   " + " ".join(f"{med[l] / med[libs[0]]:6.3f}" for l in libs[1:]) if len(libs) > 1 else ""
        print(f"{s[0]:9s} {s[1]:6d} x {s[2]:5d} x {s[3]:5d} ({s[4]})  {cells}{ratio}")
