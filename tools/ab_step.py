"""Same-box A/B of two library builds: python tools/ab_step.py ab/lib_a.so ab/lib_b.so  (each timed in its own process)."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "multimodal-sam-adapter_amd", "mmsa", "libmmsa_hip.so")
for rnd in range(2):
    for lib in sys.argv[1:]:
        shutil.copy(os.path.join(ROOT, lib), dst)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--no-extras"] + (["--no-head"] if os.environ.get("AB_NO_HEAD", "1") == "1" else []),
                             capture_output=True, text=True).stdout.strip().splitlines()[-1]
        import json
        d = json.loads(out)
        print(f"{lib}: {d['ms_per_step']:.3f} ms/step  {d['value']:.2f} img/s", flush=True)
