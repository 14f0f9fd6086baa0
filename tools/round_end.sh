#!/bin/bash
# Round-end measurement on the GPU box: GPU tests, bench line, rocprofv3 kernel stats of the same command.
# bash tools/round_end.sh <tag>   -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_tests.txt
TAG=${1:-r01_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/${TAG}_tests.txt
cat gpurun_out/${TAG}_tests.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
python tools/kstats.py gpurun_out/${TAG}_kernel_stats.csv 24 10
