#!/bin/bash
# round 5, job r: rocprofv3 kernel stats of the one-chain step on the current tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_r; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --chains 1 > $O/bench_under_rocprof.txt 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/prof
python tools/kstats.py $O/kernel_stats.csv auto 70 > $O/kstats.txt; head -60 $O/kstats.txt
