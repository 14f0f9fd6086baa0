#!/bin/bash
# round 5, job x: h8c kernel without the drain at the last pair of a tile (the epilogue waits for the next tile's first pieces together with its own vectors)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_x; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py -m gpu -x -q > $O/t1.txt 2>&1; tail -n 3 $O/t1.txt
timeout -k 10 700 python tools/gemm_sites.py --rounds 5 --only lin1,qkv,lin2,proj,extout,ffnfc1,injval,msdaoa,injout ab/libmmsa_drain0.so ab/libmmsa_new.so > $O/sites.txt 2>&1; cat $O/sites.txt
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_drain0.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
timeout -k 10 900 python -m pytest tests/test_backbone_gpu.py tests/test_inference_gpu.py -m gpu -x -q > $O/t2.txt 2>&1; tail -n 3 $O/t2.txt
