"""LayerNorm micro-benchmark on the model's shapes against a plain device copy of the same bytes: python tools/ln_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
ops = mmsa.ops
dev = "cuda:0"


def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


for rows, C in ((8192, 1024), (43008, 1024), (16384, 384), (262144, 96)):
    # several independent buffer sets so that the data does not sit in the 256 MiB Infinity Cache between iterations
    nset = max(1, int(600e6 // (rows * C * 8)))
    xs = [torch.randn(rows, C, device=dev) for _ in range(nset)]
    w = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    outs = {k: [ops.alloc_planes(rows, C, dev, fmt=f) for _ in range(nset)] for k, f in (("b3", ops.FMT_B3), ("h8", ops.FMT_H8), ("f3", ops.FMT_F3))}
    if C % 64 == 0:
        outs["h8c"] = [ops.alloc_planes(rows, C, dev, fmt=ops.FMT_H8C) for _ in range(nset)]
    ys = [torch.empty(rows, C, device=dev) for _ in range(nset)]
    it = [0]
    def nxt():
        it[0] = (it[0] + 1) % nset
        return it[0]
    res = []
    res.append(("copy", t(lambda: ys[nxt()].copy_(xs[it[0]]))))
    res.append(("LN->fp32", t(lambda: ops.layernorm(xs[nxt()], w, b, 1e-6, out=ys[it[0]]))))
    res.append(("LN->b3", t(lambda: ops.layernorm(xs[nxt()], w, b, 1e-6, out_planes=outs["b3"][it[0]]))))
    res.append(("LN->h8", t(lambda: ops.layernorm(xs[nxt()], w, b, 1e-6, out_planes=outs["h8"][it[0]]))))
    res.append(("LN->f3", t(lambda: ops.layernorm(xs[nxt()], w, b, 1e-6, out_planes=outs["f3"][it[0]]))))
    nbytes = rows * C * 8
    line = " | ".join(f"{k} {us:6.1f} us {nbytes / us / 1e6:5.2f} TB/s" for k, us in res)
    if "h8c" in outs:   # 3 bytes per element out: 7 bytes per element moved
        us = t(lambda: ops.layernorm(xs[nxt()], w, b, 1e-6, out_planes=outs["h8c"][it[0]]))
        line += f" | LN->h8c {us:6.1f} us {rows * C * 7 / us / 1e6:5.2f} TB/s"
    print(f"rows {rows:6d} C {C:4d} ({nset} buffer sets): " + line, flush=True)
