"""gemm_v3 (epilogue inside the next tile's k loop) against gemm_v2 on the GPU box.

python tools/v3_check.py check      -> bitwise comparison v3 == v2 (and fp64 sanity) over shapes / epilogue kinds / formats, several runs
python tools/v3_check.py time [dbg] -> us per launch, v2 (flavour 8) and v3 (flavour 3) interleaved in ONE process, model shapes,
                                       bf16 hi/lo and h8 operands; `dbg` = MMSA_GEMM_DEBUG of a child process (0 full, 2 no epilogue)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))

DEV = "cuda:0"

# (label, M, N, K, batch, act, out, resid, scale)
SHAPES = [
    ("lin1", 8192, 4096, 1024, 1, "gelu", "P", 0, 0), ("lin2", 8192, 1024, 4096, 1, "none", "C", 1, 0),
    ("qkv", 8192, 3072, 1024, 1, "none", "P", 0, 0), ("proj", 8192, 1024, 1024, 1, "none", "C", 1, 0),
    ("cnx2pw1", 8192, 1536, 384, 2, "gelu", "P", 0, 0), ("cnx2pw2", 8192, 384, 1536, 2, "none", "C", 1, 1),
    ("extout", 43008, 1024, 512, 1, "none", "C", 1, 0), ("ffnfc2", 43008, 1024, 256, 1, "none", "C", 1, 0),
    ("ffnfc1", 43008, 256, 1024, 1, "none", "C", 0, 0), ("val", 43008, 512, 1024, 1, "none", "C", 0, 0),
    ("cnx1pw1", 32768, 768, 192, 2, "gelu", "P", 0, 0), ("cnx3pw1", 2048, 3072, 768, 2, "gelu", "P", 0, 0),
    ("cnx3pw2", 2048, 768, 3072, 2, "none", "C", 1, 1),
    ("lin1c", 4096, 4096, 1024, 1, "gelu", "P", 0, 0), ("lin2c", 4096, 1024, 4096, 1, "none", "C", 1, 0),
]


def _setup(ops, torch, M, N, K, b, act, outk, resid, scale, h8, seed=0):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    fmt = ops.FMT_H8 if h8 else ops.FMT_B3
    af = torch.randn(b * M, K, device=DEV, generator=gen)
    wf = torch.randn(b * N, K, device=DEV, generator=gen) / K ** 0.5
    a = ops.split_planes(af, kpad=K, fmt=fmt)
    w = ops.split_planes(wf, fmt=fmt, weight=h8)
    w = ops.Planes(w.p, N, K, w.kpad, fmt, h8)
    bias = torch.randn(b * N, device=DEV, generator=gen)
    kw = dict(batch=b, m=M, stride_a=M * 2 * a.kpad, stride_w=N * 2 * w.kpad, stride_bias=N, bias=bias, act=act)
    if scale:
        kw["colscale"] = torch.rand(b * N, device=DEV, generator=gen) + 0.5
    r = torch.randn(b * M, N, device=DEV, generator=gen) if resid else None
    return a, w, kw, r, af, wf, bias


def _run(ops, torch, a, w, kw, r, M, N, b, outk, fmt, inplace=False):
    if outk == "P":
        op = ops.alloc_planes(b * M, N, DEV, fmt=fmt)
        op.p.fill_(0x7fc0)   # poison
        ops.gemm(a, w, out_planes=op, stride_cp=M * 2 * op.kpad, **kw)
        return op.p
    c = torch.full((b * M, N), float("nan"), device=DEV)
    if r is not None:
        if inplace:
            c.copy_(r)
            ops.gemm(a, w, out=c, stride_c=M * N, resid=c, stride_r=M * N, **kw)
        else:
            ops.gemm(a, w, out=c, stride_c=M * N, resid=r, stride_r=M * N, **kw)
    else:
        ops.gemm(a, w, out=c, stride_c=M * N, **kw)
    return c


def check():
    import torch
    import mmsa
    from mmsa import lib
    ops = mmsa.ops
    bad = 0
    cases = []
    for h8 in (False, True):
        for (label, M, N, K, b, act, outk, resid, scale) in SHAPES:
            cases.append((label, M, N, K, b, act, outk, resid, scale, h8))
    # small / odd tile counts, single tile per workgroup, short K, more tiles than CUs with a ragged last round
    extra = [("one", 256, 128, 128, 1, "none", "C", 1, 0), ("tiny", 512, 256, 128, 1, "gelu", "P", 0, 0), ("k64x3", 1024, 384, 192, 1, "none", "C", 0, 1),
             ("ragged", 256 * 37, 128 * 9, 320, 1, "none", "P", 0, 0), ("b3", 768, 256, 256, 3, "none", "C", 1, 1), ("deep", 512, 128, 8192, 1, "none", "C", 1, 0)]
    for e in extra:
        for h8 in (False, True):
            cases.append(e + (h8,))
    for (label, M, N, K, b, act, outk, resid, scale, h8) in cases:
        fmt = ops.FMT_H8 if h8 else ops.FMT_B3
        a, w, kw, r, af, wf, bias = _setup(ops, torch, M, N, K, b, act, outk, resid, scale, h8, seed=M + N + K)
        outs = {}
        try:
            for fl in (8, 3, 3, 3):
                lib.call("mmsa_debug_gemm_flavour", fl)
                o = _run(ops, torch, a, w, kw, r, M, N, b, outk, fmt, inplace=(fl == 3 and resid)).clone()
                torch.cuda.synchronize()
                outs.setdefault(fl, []).append(o)
        finally:
            lib.call("mmsa_debug_gemm_flavour", 0)
        ref = outs[8][0]
        ok = all(torch.equal(ref, o) for o in outs[3])
        finite = bool(torch.isfinite(ref.float()).all()) if outk == "C" else True
        # fp64 sanity of v2 itself on the first batch element (planes: skip)
        err = float("nan")
        if outk == "C":
            x = af[:M].double() @ wf[:N].double().t() + bias[:N].double()
            if scale:
                x = x * kw["colscale"][:N].double()
            if r is not None:
                x = x + r[:M].double()
            err = ((ref[:M].double() - x).norm() / x.norm()).item()
        nd = 0 if ok else int(sum((ref != o).sum().item() for o in outs[3]))
        print(f"{'h8' if h8 else 'b3'} {label:8s} M={M:6d} N={N:5d} K={K:5d} b={b} {act:4s} {outk} res={resid} sc={scale}: "
              f"{'BITWISE OK' if ok else 'DIFF ' + str(nd)}  finite={finite} v2-vs-fp64 {err:.2e}", flush=True)
        bad += 0 if ok else 1
    print("FAILED" if bad else "ALL OK", bad)
    return bad


FLAVOURS = tuple(int(v) for v in os.environ.get("V3_CHECK_FLAVOURS", "8,3").split(","))   # mmsa_debug_gemm_flavour values timed against each other


def time_worker():
    import torch
    import mmsa
    from mmsa import lib
    ops = mmsa.ops
    print(f"{'':12s}" + " ".join(f"{s[0]:>8s}" for s in SHAPES))
    for h8 in ((False,) if 4 in FLAVOURS else (False, True)):
        rows = {fl: [] for fl in FLAVOURS}
        for (label, M, N, K, b, act, outk, resid, scale) in SHAPES:
            fmt = ops.FMT_H8 if h8 else ops.FMT_B3
            a, w, kw, r, *_ = _setup(ops, torch, M, N, K, b, act, outk, resid, scale, h8)
            if outk == "P":
                op = ops.alloc_planes(b * M, N, DEV, fmt=fmt)
                call = lambda: ops.gemm(a, w, out_planes=op, stride_cp=M * 2 * op.kpad, **kw)
            else:
                c = torch.randn(b * M, N, device=DEV)
                if resid:
                    call = lambda: ops.gemm(a, w, out=c, stride_c=M * N, resid=c, stride_r=M * N, beta=0.5, **kw)   # in place like the model (beta keeps it bounded)
                else:
                    call = lambda: ops.gemm(a, w, out=c, stride_c=M * N, **kw)
            best = {fl: 1e9 for fl in FLAVOURS}
            for rnd in range(3):          # interleaved rounds, one process
                for fl in FLAVOURS:
                    lib.call("mmsa_debug_gemm_flavour", fl)
                    for _ in range(3):
                        call()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    reps = 20
                    for _ in range(reps):
                        call()
                    torch.cuda.synchronize()
                    best[fl] = min(best[fl], (time.perf_counter() - t0) / reps * 1e6)
            lib.call("mmsa_debug_gemm_flavour", 0)
            for fl in FLAVOURS:
                rows[fl].append(best[fl])
        tag = "h8" if h8 else "b3"
        flops = [2.0 * s[1] * s[2] * s[3] * s[4] for s in SHAPES]
        for fl in FLAVOURS:
            print(f"{tag} fl{fl} us   " + " ".join(f"{u:8.1f}" for u in rows[fl]))
        a_, b_ = FLAVOURS[0], FLAVOURS[1]
        print(f"{tag} fl{b_}/fl{a_}   " + " ".join(f"{x / y:8.3f}" for x, y in zip(rows[b_], rows[a_])))
        print(f"{tag} fl{b_} TF/s " + " ".join(f"{f / u / 1e6:8.1f}" for f, u in zip(flops, rows[b_])), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(1 if check() else 0)
    if mode == "time_worker":
        time_worker()
        sys.exit(0)
    for dbg in (sys.argv[2:] or ["0", "2"]):
        print(f"== MMSA_GEMM_DEBUG={dbg}", flush=True)
        env = dict(os.environ, MMSA_GEMM_DEBUG=dbg)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "time_worker"], env=env, capture_output=True, text=True)
        print(out.stdout[-6000:] if out.stdout else out.stderr[-3000:], flush=True)
