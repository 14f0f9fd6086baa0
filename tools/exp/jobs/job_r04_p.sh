#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_p; mkdir -p $O
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-verify > $O/bench$i.json 2> $O/bench$i.err; done
MMSA_GEMM_FLAVOUR=8 timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-verify > $O/bench_fl8.json 2> $O/bench_fl8.err
python - <<'PY'
import json
for n in ("bench1","bench2","bench_fl8"):
    j=json.loads(open(f'gpurun_out/r04_p/{n}.json').read().strip().splitlines()[-1])
    print(n, "value", j["value"], "chains", j["config"]["chains_per_gpu"], j["chains_probe_ms"], "replay median", j["replay_ms"]["median"], "enc", j["encoder_only"]["value"], "worst", (j["worst_case_precision"] or {}).get("value"))
PY
