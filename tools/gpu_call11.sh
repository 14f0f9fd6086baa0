#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11
L=multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
for v in base mx; do
  cp ab/lib_$v.so $L
  echo "## $v" | tee -a gpurun_out/c11/mx.txt
  timeout 300 python tools/gemm_ablate.py 0 2 2>&1 | tee -a gpurun_out/c11/mx.txt
done
echo "## stamp_mx lin2" | tee -a gpurun_out/c11/stamps.txt
timeout 200 python tools/gemm_stamps.py ab/lib_stamp_mx.so 8192 1024 4096 2>&1 | grep "wave [04] " | tee -a gpurun_out/c11/stamps.txt
cp ab/lib_base.so $L
