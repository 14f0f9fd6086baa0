#!/bin/bash
# round 5, job v: what the lin1 epilogue's time is made of -- instruction issue costs (tools/exp/valu_rate.hip) and lin1 without GELU / without the row-normalising form
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_v; mkdir -p $O
timeout -k 10 60 tools/exp/bin/valu_rate > $O/valu.txt 2>&1; cat $O/valu.txt
timeout -k 10 600 python tools/gemm_sites.py --rounds 3 --only lin1,lin1none,lin1norn,lin1bare,qkv multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_knobs.so:MMSA_GEMM_DEBUG=2 > $O/sites.txt 2>&1; cat $O/sites.txt
