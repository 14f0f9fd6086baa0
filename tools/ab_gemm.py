"""Same-box A/B of library builds on the GEMM micro-benchmark: python tools/ab_gemm.py ab/lib_a.so ab/lib_b.so ..."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "multimodal-sam-adapter_amd", "mmsa", "libmmsa_hip.so")
for rnd in range(2):
    for lib in sys.argv[1:]:
        shutil.copy(os.path.join(ROOT, lib), dst)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_bench.py")], capture_output=True, text=True).stdout
        short = " ".join(f"{l.split('TFLOP/s')[0].split()[-1]:>6s}" for l in out.splitlines() if "TFLOP/s" in l)
        print(f"{lib:18s} {short}", flush=True)
