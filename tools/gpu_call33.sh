#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c33
timeout 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | grep -v "^E    .*tensor(\[" | tail -5 | tee gpurun_out/c33/tests.txt
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so /tmp/keep.so
for v in pairstore direct; do
  cp ab/lib_$v.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
  echo "## $v" | tee -a gpurun_out/c33/ablate.txt
  MMSA_ABLATE_FMT=h8 timeout 300 python tools/gemm_ablate.py 0 2>&1 | tee -a gpurun_out/c33/ablate.txt
done
AB_NO_HEAD=0 timeout 800 python tools/ab_step.py ab/lib_pairstore.so ab/lib_direct.so 2>&1 | grep ms/step | tee gpurun_out/c33/ab.txt
cp /tmp/keep.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
