#!/bin/bash
# round 5, job ah: per-value epilogue arithmetic in packed pairs (bias / row norm / scale / residual): parity (bit-identity with the staged path), sites, step
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_ah; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py -m gpu -x -q > $O/t1.txt 2>&1; tail -n 3 $O/t1.txt
timeout -k 10 700 python tools/gemm_sites.py --rounds 5 --only lin1,qkv,lin2,proj,cnx2pw1,injout ab/libmmsa_prev.so ab/libmmsa_new.so > $O/sites.txt 2>&1; cat $O/sites.txt
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_prev.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
