#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c22
timeout 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | grep -v "^E    .*tensor(\[" | tail -4 | tee gpurun_out/c22/tests.txt
for rm in 0 1 0 1; do
  echo "## rowmajor=$rm" | tee -a gpurun_out/c22/ablate.txt
  if [ $rm = 1 ]; then export MMSA_GEMM_ROWMAJOR=1; else unset MMSA_GEMM_ROWMAJOR; fi
  MMSA_ABLATE_FMT=h8 timeout 300 python tools/gemm_ablate.py 0 2>&1 | tee -a gpurun_out/c22/ablate.txt
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench rowmajor=$rm', d['value'], d['ms_per_step'], d['encoder_only']['ms_per_step'])" | tee -a gpurun_out/c22/ablate.txt
done
unset MMSA_GEMM_ROWMAJOR
bash tools/pmc_traffic.sh r02 2>&1 | grep -E "per_launch|rc="
cp profiles/r02_gemm_traffic.json gpurun_out/c22/
rm -rf gpurun_out/pmc_r02_*
