#!/bin/bash
# round 5, job af: epilogue takes the tile coordinates from the operand cursor (no integer divisions per tile boundary): parity, stamps, sites
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_af; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py -m gpu -x -q -k "gemm or fold or h8c or convnext or plane" > $O/t1.txt 2>&1; tail -n 3 $O/t1.txt
timeout -k 10 300 python tools/epi_stamps.py ab/libmmsa_estamp.so lin1 qkv extout > $O/stamps.txt 2>&1; grep -v amdgpu $O/stamps.txt | grep -A3 "^lin1\|^qkv\|^extout"
timeout -k 10 700 python tools/gemm_sites.py --rounds 5 --only lin1,qkv,lin2,proj,extout,ffnfc1,injval,msdaoa,injout ab/libmmsa_coords0.so ab/libmmsa_new.so > $O/sites.txt 2>&1; cat $O/sites.txt
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_coords0.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
