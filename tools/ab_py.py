"""Same-box A/B of two versions of a Python file: python tools/ab_py.py <target path> <version a> <version b>."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
target = os.path.join(ROOT, sys.argv[1])
for rnd in range(2):
    for v in sys.argv[2:]:
        shutil.copy(os.path.join(ROOT, v), target)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--no-head"],
                             capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(f"{v}: {d['ms_per_step']:.3f} ms/step  {d['value']:.2f} img/s", flush=True)
