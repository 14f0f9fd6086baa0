#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
timeout 300 python tools/ln_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c9/ln.txt
for dbg in 0 2 3; do
for cfg in "0 0" "64 0" "0 64" "64 64" "192 192"; do
  set -- $cfg
  MMSA_GEMM_DEBUG=$dbg MMSA_GEMM_WPAD=$2 timeout 200 python tools/gemm_pad_exp.py $1 2>&1 | grep apad | tee -a gpurun_out/c9/pad.txt
done
done
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | tail -3
