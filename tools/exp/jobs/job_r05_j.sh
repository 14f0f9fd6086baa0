#!/bin/bash
# round 5, job j: same-box A/B of the step: staged epilogue only (-DMMSA_EPI_REGS=0, otherwise the current tree) | current tree
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_j; mkdir -p $O
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_new.so
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_regs0.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
