#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c26
for cap in 256 128 160 192; do
  MMSA_GEMM_MAX_GRID=$cap timeout 300 python tools/split_batch_bench.py 1 2 2>&1 | grep "images/s" | tee -a gpurun_out/c26/split.txt
done
MMSA_GEMM_MAX_GRID=256 timeout 300 python tools/split_batch_bench.py 2 1 2>&1 | grep "images/s" | tee -a gpurun_out/c26/split.txt
MMSA_GEMM_MAX_GRID=128 timeout 300 python tools/split_batch_bench.py 2 2 2>&1 | grep "images/s" | tee -a gpurun_out/c26/split.txt
