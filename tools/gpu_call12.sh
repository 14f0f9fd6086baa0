#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12
timeout 900 python -m pytest tests/test_planes_gpu.py -x -q -m gpu 2>&1 | grep -v "^E    .*tensor(\[" | tail -25 > gpurun_out/c12/tests_planes.txt
cat gpurun_out/c12/tests_planes.txt
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_planes_gpu.py 2>&1 | grep -v "^E    .*tensor(\[" | tail -25 > gpurun_out/c12/tests.txt
cat gpurun_out/c12/tests.txt
for h8 in vit none; do
  MMSA_H8=$h8 timeout 300 python bench.py --no-cpu-baseline --steps 20 2>gpurun_out/c12/bench_$h8.err | tail -1 > gpurun_out/c12/bench_$h8.json
  python -c "import json,sys; d=json.load(open('gpurun_out/c12/bench_$h8.json')); print('$h8', d['value'], d['ms_per_step'], d['encoder_only'], d['verified'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_step'])" || tail -5 gpurun_out/c12/bench_$h8.err
done
