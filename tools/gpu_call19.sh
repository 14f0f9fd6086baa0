#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c19
for r in 4 2 1 4 2 1; do
  MMSA_LN_ROWS=$r timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('LN_ROWS=$r', d['value'], d['ms_per_step'], d['encoder_only']['ms_per_step'])" | tee -a gpurun_out/c19/ln_rows.txt
done
