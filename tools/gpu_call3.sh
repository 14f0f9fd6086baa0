#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c3/tests.txt
cat gpurun_out/c3/tests.txt
timeout 600 python bench.py --config vith1024 --no-cpu-baseline --steps 10 --warmup 3 2> gpurun_out/c3/vith.err | tail -1 > gpurun_out/c3/vith.json
cat gpurun_out/c3/vith.json; tail -5 gpurun_out/c3/vith.err
