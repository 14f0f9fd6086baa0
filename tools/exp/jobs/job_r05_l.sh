#!/bin/bash
# round 5, job l: clamp watch, second form (stateless check in the staged path, running maximum in the register path only, residual rows one sub-tile ahead)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_l; mkdir -p $O
timeout -k 10 900 python tools/gemm_sites.py --rounds 5 --only lin1,qkv,lin2,proj,cnx2pw1,injout,extout,ffnfc2 ab/libmmsa_regs0.so ab/libmmsa_noclamp.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so > $O/sites.txt 2>&1; cat $O/sites.txt
for v in regs0 noclamp; do MMSA_LIB=$PWD/ab/libmmsa_$v.so timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_$v.txt 2>&1; echo "$v: $(sed -n 2p $O/shapes_$v.txt)"; done
timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_new.txt 2>&1; echo "new: $(sed -n 2p $O/shapes_new.txt)"
timeout -k 10 300 python -m pytest tests/test_backbone_gpu.py tests/test_planes_gpu.py -m gpu -x -q -k "clamp_flag or register_epilogue" > $O/t.txt 2>&1; tail -n 2 $O/t.txt
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_new.so
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_regs0.so ab/libmmsa_noclamp.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
