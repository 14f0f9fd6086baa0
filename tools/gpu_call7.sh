#!/bin/bash
# session-2 baseline of round 2: GPU tests, bench line, rocprof kernel stats, per-shape GEMM table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c7/tests.txt
cat gpurun_out/c7/tests.txt
timeout 600 python bench.py 2> gpurun_out/c7/bench.err | tail -1 > gpurun_out/c7/bench.json
cat gpurun_out/c7/bench.json; tail -3 gpurun_out/c7/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c7/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(ls gpurun_out/c7/prof/*/*kernel_stats.csv | head -1) gpurun_out/c7/kernel_stats.csv
rm -rf gpurun_out/c7/prof
python tools/kstats.py gpurun_out/c7/kernel_stats.csv 24 45 | tee gpurun_out/c7/kstats.txt
timeout 300 python tools/gemm_shapes.py > gpurun_out/c7/shapes.txt 2>&1
cat gpurun_out/c7/shapes.txt
