#!/bin/bash
# round 5, job u: GELU forms in the GEMM epilogue (common.h MMSA_GELU_FORM: 0 packed pairs, 1 single values with operand modifiers, 2 two pairs in lockstep)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_u; mkdir -p $O
timeout -k 10 700 python tools/gemm_sites.py --rounds 5 --only lin1,cnx2pw1,cnx1pw1,cnx3pw1 ab/libmmsa_gelu0.so ab/libmmsa_gelu1.so ab/libmmsa_gelu2.so > $O/sites.txt 2>&1; cat $O/sites.txt
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_gelu0.so ab/libmmsa_gelu1.so ab/libmmsa_gelu2.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_gelu0.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
