#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_v3; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/tests.txt 2>&1; tail -14 $O/tests.txt
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify > $O/bench$i.json 2> $O/bench$i.err; done
python - <<'PY'
import json
for n in ("bench1","bench2"):
    j=json.loads(open(f'gpurun_out/r04_v3/{n}.json').read().strip().splitlines()[-1])
    r=j["roofline"]
    print(n, "value", j["value"], "chains", j["config"]["chains_per_gpu"], j["chains_probe_ms"], "replay median", j["replay_ms"]["median"], "enc", j["encoder_only"]["value"], "frac", r["frac"], r["achieved"], r["kernel_ms_per_step"], "worst", (j.get("worst_case_precision") or {}).get("value"))
PY
