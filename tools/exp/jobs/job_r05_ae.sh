#!/bin/bash
# round 5, job ae: s_setprio 1 around the matrix parts of the h8c kernel's pairs (the two waves of a SIMD are in opposite phases)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_ae; mkdir -p $O
timeout -k 10 700 python tools/gemm_sites.py --rounds 5 --only lin1,qkv,lin2,proj,extout,ffnfc1,injval multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_prio1.so > $O/sites.txt 2>&1; cat $O/sites.txt
