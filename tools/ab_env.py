"""Same-box A/B of environment settings: each variant is bench.py in its own process, variants interleaved over several rounds.
    python tools/ab_env.py [--rounds 2] [--steps 20] [--verify] name1:VAR=val+VAR2=val name2: name3@fold_adapter_ln=False+h8c=False ...
(`name:ENV=..` sets environment variables -- MMSA_LIB to time another library build --, `name@attr=value+..` passes `--set attr=value` to bench.py: a
backbone attribute; an empty setting list = the default build).  Prints ms/step per run and the per-variant minimum; with --verify also the golden probe
error of the timed path (`verified.golden_max_rel`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, steps, verify = 2, 20, False
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    elif args[0] == "--steps":
        steps = int(args[1]); args = args[2:]
    elif args[0] == "--verify":
        verify = True; args = args[1:]
    else:
        raise SystemExit(f"unknown flag {args[0]}")
variants = []
for a in args:
    a, _, attrs = a.partition("@")
    name, _, sets = a.partition(":")
    env = dict(kv.split("=", 1) for kv in sets.split("+") if kv)
    variants.append((name, env, [x for kv in attrs.split("+") if kv for x in ("--set", kv)]))
best = {}
for rnd in range(rounds):
    for name, env, extra in variants:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--no-extras"] + extra
        if not verify:
            cmd.append("--no-verify")
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env))
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print(f"{name}: FAILED\n{r.stderr[-1500:]}", flush=True)
            continue
        extra = ""
        if verify and isinstance(d.get("verified"), dict):
            extra = "  " + " ".join(f"{k}={v}" for k, v in d["verified"].items() if "max_rel" in k or "golden_probes" in k)
        print(f"round {rnd} {name:14s} {d['ms_per_step']:.3f} ms/step {d['value']:.2f} img/s  encoder-only {d.get('encoder_only')}{extra}", flush=True)
        best[name] = min(best.get(name, 1e9), d["ms_per_step"])
print("best: " + "  ".join(f"{k} {v:.3f}" for k, v in best.items()))
