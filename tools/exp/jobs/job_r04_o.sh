#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_o; mkdir -p $O
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 pair_old:MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_pairold.so+MMSA_DWPAIR_STRIP=0 pair_strip: chains2:MMSA_BENCH_FORCE_CHAINS=1 > $O/ab.txt 2>&1
cat $O/ab.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-verify > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04_o/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "chains", j["config"]["chains_per_gpu"], j["chains_probe_ms"], "replay", j["replay_ms"]["median"], "worst", j["worst_case_precision"])
PY
