#!/bin/bash
# round 6, end c: the rocprofv3 kernel stats of the default step alone (one chain, no extra legs: the ViT-H / frame / eager legs had been in the trace of end b)
TAG=r06_z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --chains 1 > gpurun_out/$TAG/bench_under_rocprof.txt 2>&1
cp $(ls gpurun_out/$TAG/prof/*/*kernel_stats.csv | head -1) gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/prof
python tools/kstats.py gpurun_out/$TAG/kernel_stats.csv auto 60 > gpurun_out/$TAG/kstats.txt; head -16 gpurun_out/$TAG/kstats.txt
tail -1 gpurun_out/$TAG/bench_under_rocprof.txt | cut -c1-300
