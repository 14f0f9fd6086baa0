"""Same-box A/B of environment settings on the bench step: python tools/ab_env.py MMSA_GEMM_PP=0 MMSA_GEMM_PP=1"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for kv in sys.argv[1:]:
        env = dict(os.environ); k, v = kv.split("="); env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-roofline"] + (["--no-head"] if os.environ.get("AB_NO_HEAD", "1") == "1" else []),
                             capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(f"{kv}: {d['ms_per_step']:.3f} ms/step  {d['value']:.2f} img/s", flush=True)
