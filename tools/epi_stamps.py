"""Where a tile boundary of the h8c GEMM spends its time: shader-clock stamps of workgroup 0's second tile boundary (library built with -DHC_EPI_STAMP:
tools/build_variant.sh ab/libmmsa_estamp.so gemm_h8c.hip -DHC_EPI_STAMP).   python tools/epi_stamps.py ab/libmmsa_estamp.so [site ...]
Sites as in tools/gemm_sites.py (lin1, qkv, proj, extout, ...).  Per wave: cycles from the k loop's last barrier to: vectors requested | arrived | sub-tile 0..3 done |
epilogue left | barrier behind it passed."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMSA_LIB"] = os.path.join(ROOT, sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import mmsa  # noqa: E402
from mmsa import lib  # noqa: E402
import gemm_sites  # noqa: E402

ops = mmsa.ops
dev = "cuda:0"
FM = {"b3": ops.FMT_B3, "h8": ops.FMT_H8, "h8c": ops.FMT_H8C, "f3": ops.FMT_F3}
want = sys.argv[2:] or ["lin1"]
h = ctypes.CDLL(os.environ["MMSA_LIB"])
for (label, M, N, K, b, fmt_n, act, outk, pf, resid, rs, rn, cs) in gemm_sites.SITES:
    if label not in want:
        continue
    fmt = FM[fmt_n]
    a = ops.split_planes(torch.randn(b * M, K, device=dev), kpad=K, fmt=fmt)
    wa = ops.split_planes(torch.randn(b * N, K, device=dev) / K ** 0.5, fmt=fmt)
    w = ops.Planes(wa.p, N, K, wa.kpad, fmt, False)
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
    for _ in range(5):
        ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    rc = h.mmsa_debug_epi_stamps(buf)
    print(f"{label}: M {M} N {N} K {K} act {act} out {outk} resid {resid} strip sums {rs} row norm {rn}   (rc {rc}); cycles after the k loop's last barrier")
    print("  wave   requested   arrived   sub0   sub1   sub2   sub3   left   barrier")
    for wv in range(8):
        t = [buf[wv * 16 + q] for q in range(9)]
        if not t[0]:
            continue
        print(f"  {wv:4d} " + " ".join(f"{(x - t[0]) if x else -1:8d}" for x in t[1:]))
