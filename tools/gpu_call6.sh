#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
timeout 600 python -m pytest tests/test_inference_gpu.py -x -q -m gpu 2>&1 | tail -4
MMSA_GEMM_NO_TINY=1 timeout 600 python -m pytest tests/test_inference_gpu.py -x -q -m gpu -k muses 2>&1 | tail -4
for cap in 256 128 144 176; do
  MMSA_GEMM_MAX_GRID=$cap timeout 300 python tools/split_batch_bench.py 1 2 2>&1 | grep "images/s" | tee -a gpurun_out/c6/split.txt
done
MMSA_GEMM_MAX_GRID=128 timeout 300 python tools/split_batch_bench.py 2 2 2>&1 | grep "images/s" | tee -a gpurun_out/c6/split.txt
