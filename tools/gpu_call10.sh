#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
for v in stamp stamp_mx; do
  echo "## $v lin2" | tee -a gpurun_out/c10/stamps.txt
  timeout 200 python tools/gemm_stamps.py ab/lib_$v.so 8192 1024 4096 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c10/stamps.txt
done
cp ab/lib_base.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
