"""Timing experiment: per-phase cycle stamps of the ping-pong GEMM main loop (library built with -DV2_STAMP).
python tools/gemm_stamps.py ab/lib_stamp.so M N K"""
import ctypes, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shutil.copy(os.path.join(ROOT, sys.argv[1]), os.path.join(ROOT, "multimodal-sam-adapter_amd", "mmsa", "libmmsa_hip.so"))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch  # noqa: E402
import mmsa  # noqa: E402
from mmsa import lib  # noqa: E402
ops = mmsa.ops
M, N, K = (int(v) for v in sys.argv[2:5])
h8 = os.environ.get("MMSA_ABLATE_FMT") == "h8"
fmt = ops.FMT_H8 if h8 else ops.FMT_B3
a = ops.split_planes(torch.randn(M, K, device="cuda:0"), kpad=K, fmt=fmt)
w = ops.split_planes(torch.randn(N, K, device="cuda:0") / K ** 0.5, fmt=fmt, weight=h8)
out = torch.empty(M, N, device="cuda:0")
for _ in range(5):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 324)()
h = lib.handle() if hasattr(lib, "handle") else lib._lib
rc = h.mmsa_debug_stamps(buf)
if os.environ.get("MMSA_ISTAMP"):   # interval stamps of the straight-line steps: clock after each barrier, waves 0 and 4
    t0 = min(buf[i] for i in range(32) if buf[i])
    for grp in range(2):
        ts = [buf[grp * 16 + i] - t0 for i in range(16)]
        print(f"group {grp}: barrier exits " + " ".join(f"{t:6d}" for t in ts))
        print(f"         intervals      " + " ".join(f"{b - a:6d}" for a, b in zip(ts, ts[1:])) + "   (even = after the read phase's barrier -> MFMA phase, odd = MFMA phase's barrier -> read phase)")
    cyc, real = buf[322] - buf[320], buf[323] - buf[321]
    print(f"workgroup 0: {cyc} shader cycles in {real / 100:.1f} us -> average shader clock {cyc / real * 0.1:.3f} GHz")
    sys.exit(0)
names = ["start", "reads issued", "dma issued", "lgkm0", "vm wait", "barrier1", "mfma issued", "vm wait", "barrier2"]
t0 = min(buf[(wv * 4) * 10] for wv in range(8))
for wv in range(8):
    for s in range(4):
        b = [(buf[(wv * 4 + s) * 10 + q] - t0) for q in range(9)]
        print(f"wave {wv} grp {wv >> 2} kt {8 + s}: start {b[0]:6d} | " + " ".join(f"{names[q + 1]} +{b[q + 1] - b[q]:4d}" for q in range(8)) + f" | step {b[8] - b[0]}")
cyc, real = buf[322] - buf[320], buf[323] - buf[321]
print(f"workgroup 0: {cyc} shader cycles in {real} ticks of the 100 MHz counter = {real / 100:.1f} us -> average shader clock {cyc / real * 0.1:.3f} GHz" if real else "no clock sample")
