#!/bin/bash
# round 5, job ac: global attention (all-fp16 form) with a second K / V buffer in the unused lo-plane regions: one barrier per key block instead of two
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py tests/test_bookkeeping_gpu.py -m gpu -x -q -k "attn or attention or global or window" > $O/t1.txt 2>&1; tail -n 3 $O/t1.txt
timeout -k 10 300 python tools/gattn_bench.py 2 ab/libmmsa_attn_db0.so ab/libmmsa_new.so > $O/gattn.txt 2>&1; cat $O/gattn.txt
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_attn_db0.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
timeout -k 10 900 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q > $O/t2.txt 2>&1; tail -n 3 $O/t2.txt
