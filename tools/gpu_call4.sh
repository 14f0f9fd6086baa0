#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
timeout 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_bookkeeping_gpu.py tests/test_inference_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c4/tests.txt
cat gpurun_out/c4/tests.txt
MMSA_GEMM_NW=8 timeout 300 python tools/gemm_ablate.py 0 2>&1 | tee gpurun_out/c4/ablate_nw8.txt
MMSA_GEMM_NW=4 timeout 300 python tools/gemm_ablate.py 0 2>&1 | tee gpurun_out/c4/ablate_nw4.txt
for k in 0 512 1024 4096; do
  echo "NW4_MAXK=$k"
  MMSA_GEMM_NW4_MAXK=$k timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['encoder_only'], d['verified'])"
done 2>&1 | tee gpurun_out/c4/maxk.txt
