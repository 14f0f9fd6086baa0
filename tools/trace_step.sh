#!/bin/bash
# kernel trace of graph-replayed steps + concurrency / idle-gap analysis:  bash tools/trace_step.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_step
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
f=$(ls gpurun_out/trace_step/*/*kernel_trace.csv | head -1)
python tools/trace_critical.py $f
