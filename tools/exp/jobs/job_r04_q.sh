#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_q; mkdir -p $O
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-verify > $O/bench$i.json 2> $O/bench$i.err; done
python - <<'PY'
import json
for n in ("bench1","bench2"):
    j=json.loads(open(f'gpurun_out/r04_q/{n}.json').read().strip().splitlines()[-1])
    print(n, "value", j["value"], "chains", j["config"]["chains_per_gpu"], j["chains_probe_ms"], "replay median", j["replay_ms"]["median"], "enc", j["encoder_only"]["value"])
PY
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --chains 1 > /dev/null 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv && rm -rf $O/prof
python tools/kstats.py $O/kernel_stats.csv 62 12 > $O/kstats.txt; head -50 $O/kstats.txt
bash tools/pmc_traffic.sh r04q > $O/pmc.txt 2>&1; tail -5 $O/pmc.txt
cp profiles/r04q_hbm_kernels.json profiles/r04q_gemm_traffic.json $O/
rm -rf gpurun_out/pmc_r04q_*
