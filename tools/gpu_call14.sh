#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c14
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^E    .*tensor(\[" | tail -15 > gpurun_out/c14/tests.txt
cat gpurun_out/c14/tests.txt
for h8 in vit,inter,up vit none; do
  MMSA_H8=$h8 timeout 300 python bench.py --no-cpu-baseline --steps 20 2>gpurun_out/c14/bench_$h8.err | tail -1 > gpurun_out/c14/bench_$h8.json
  python -c "import json,sys; d=json.load(open('gpurun_out/c14/bench_$h8.json')); print('$h8', d['value'], d['ms_per_step'], d['encoder_only'], d['verified'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_step'])" || tail -5 gpurun_out/c14/bench_$h8.err
done
