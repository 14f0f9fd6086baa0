#!/bin/bash
# round 5, job f: all sixteen residual loads of a tile up front -- previous build (one sub-tile ahead) | shipped | the same with the fp32-only register variants on
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_f; mkdir -p $O
timeout -k 10 600 python tools/gemm_sites.py --rounds 3 ab/libmmsa_nostagger.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_fp32regs.so > $O/sites.txt 2>&1; cat $O/sites.txt
