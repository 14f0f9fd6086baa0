#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c32
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^E    .*tensor(\[" | tail -5 | tee gpurun_out/c32/tests.txt
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so /tmp/keep.so
AB_NO_HEAD=0 timeout 800 python tools/ab_step.py ab/lib_prepair.so ab/lib_pairstore.so 2>&1 | grep ms/step | tee gpurun_out/c32/ab.txt
cp /tmp/keep.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
