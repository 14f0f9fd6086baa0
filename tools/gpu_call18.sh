#!/bin/bash
# tag-able measurement call: bench line, rocprof kernel stats of the same command, per-shape GEMM table
TAG=${1:-r02_a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 600 python bench.py 2> gpurun_out/$TAG/bench.err | tail -1 > gpurun_out/$TAG/bench.json
cat gpurun_out/$TAG/bench.json; tail -2 gpurun_out/$TAG/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(ls gpurun_out/$TAG/prof/*/*kernel_stats.csv | head -1) gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/prof
python tools/kstats.py gpurun_out/$TAG/kernel_stats.csv 62 40 | tee gpurun_out/$TAG/kstats.txt
timeout 300 python tools/gemm_shapes.py > gpurun_out/$TAG/shapes.txt 2>&1
head -30 gpurun_out/$TAG/shapes.txt
