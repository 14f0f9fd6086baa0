"""Same-box A/B of library builds on the model's GEMM shapes (tools/gemm_ablate.py worker): python tools/ab_ablate.py ab/lib_a.so ab/lib_b.so ...
MMSA_ABLATE_FMT=h8 / MMSA_GEMM_DEBUG are passed through."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "multimodal-sam-adapter_amd", "mmsa", "libmmsa_hip.so")
for rnd in range(2):
    for lib in sys.argv[1:]:
        shutil.copy(os.path.join(ROOT, lib), dst)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_ablate.py"), "worker"], capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
        print(f"{lib:18s} {line}", flush=True)
