#!/bin/bash
# round 5, job m: "rows" form of the fp32-only GEMM outputs -- parity, then sites and step against the same tree without it
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "gemm or fold or f3" > $O/t.txt 2>&1; tail -n 5 $O/t.txt
timeout -k 10 900 python tools/gemm_sites.py --rounds 5 --only extout,ffnfc2,ffnfc1,injval,msdaoa,injoa,extval,cnx2pw2 ab/libmmsa_norows.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so > $O/sites.txt 2>&1; cat $O/sites.txt
MMSA_LIB=$PWD/ab/libmmsa_norows.so timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_norows.txt 2>&1; echo "norows: $(sed -n 2p $O/shapes_norows.txt)"
timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_new.txt 2>&1; echo "new: $(sed -n 2p $O/shapes_new.txt)"
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_new.so
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_norows.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
