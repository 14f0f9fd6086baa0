#!/bin/bash
# timing experiment: MFMA mix of the fp16-hi + fp8-cross operand format on the unchanged data path (ab/lib_mx.so: garbage results) vs the split3 build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
L=multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
for rnd in 1 2; do
for v in base mx; do
  cp ab/lib_$v.so $L
  echo "## $v" | tee -a gpurun_out/c8/mx.txt
  timeout 300 python tools/gemm_ablate.py 0 2 2>&1 | tee -a gpurun_out/c8/mx.txt
done
done
cp ab/lib_base.so $L
