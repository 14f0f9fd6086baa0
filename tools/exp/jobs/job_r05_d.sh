#!/bin/bash
# round 5, job d: what is left of the epilogue -- shipped build | the same without the register path's global stores | k loop only (debug-knob build, MMSA_GEMM_DEBUG=2)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_d; mkdir -p $O
timeout -k 10 900 python tools/gemm_sites.py --rounds 3 multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_nostore.so ab/libmmsa_knobs.so:MMSA_GEMM_DEBUG=2 > $O/sites.txt 2>&1
cat $O/sites.txt
