#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c27
timeout 900 python -m pytest tests/test_inference_gpu.py -x -q -m gpu 2>&1 | grep -v "^E    .*tensor(\[" | tail -15 | tee gpurun_out/c27/tests.txt
for ch in 2 1 2 1; do
  timeout 400 python bench.py --no-cpu-baseline --no-roofline --chains $ch --steps 20 2>gpurun_out/c27/bench_$ch.err | tail -1 > gpurun_out/c27/bench_$ch.json
  python -c "import json,sys; d=json.load(open('gpurun_out/c27/bench_$ch.json')); print('chains=$ch', d['value'], d['ms_per_step'], d['encoder_only'], d['verified'])" || tail -8 gpurun_out/c27/bench_$ch.err
done
