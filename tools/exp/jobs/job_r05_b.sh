#!/bin/bash
# round 5, job b: stand-alone site timing of four builds of the GEMM epilogue (same box): round-4 library, staged epilogue only (+ opaque lane id in gemm_v2),
# register epilogue without its fp32-only variants, register epilogue (in-tree)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b; mkdir -p $O
timeout -k 10 900 python tools/gemm_sites.py --rounds 3 ab/libmmsa_r04.so ab/libmmsa_regs0.so ab/libmmsa_nofp32.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so > $O/sites.txt 2>&1
cat $O/sites.txt
