#!/bin/bash
# first GPU call of round 2: tests, bench line, per-shape GEMM table, SLP on/off A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c1/tests.txt
cat gpurun_out/c1/tests.txt
timeout 600 python bench.py 2> gpurun_out/c1/bench.err | tail -1 > gpurun_out/c1/bench.json
cat gpurun_out/c1/bench.json; tail -3 gpurun_out/c1/bench.err
timeout 300 python tools/gemm_shapes.py > gpurun_out/c1/shapes.txt 2>&1
cat gpurun_out/c1/shapes.txt
timeout 600 python tools/ab_step.py ab/lib_slp.so ab/lib_noslp.so 2>&1 | tail -6 | tee gpurun_out/c1/ab_slp.txt
