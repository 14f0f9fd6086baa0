#!/bin/bash
# round 5, job k: what the clamp watch and the all-up-front residual loads cost the GEMM epilogue (same box): sites stand-alone (5 rounds) and in the model
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_k; mkdir -p $O
timeout -k 10 900 python tools/gemm_sites.py --rounds 5 --only lin1,qkv,lin2,proj,cnx2pw1,injout,extout ab/libmmsa_regs0.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_noclamp.so ab/libmmsa_pf1.so ab/libmmsa_noclamp_pf1.so > $O/sites.txt 2>&1; cat $O/sites.txt
for v in regs0 noclamp pf1 noclamp_pf1; do MMSA_LIB=$PWD/ab/libmmsa_$v.so timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_$v.txt 2>&1; echo "$v: $(sed -n 2p $O/shapes_$v.txt)"; done
timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_new.txt 2>&1; echo "new: $(sed -n 2p $O/shapes_new.txt)"
