#!/bin/bash
# round 4, job h: h8c sites in the model: backbone tests, same-box A/B of the step (h8c off / on / flavour 8 forced), per-shape full-kernel timing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_backbone_gpu.py tests/test_planes_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc=$?" >> $O/tests.txt
tail -n 6 $O/tests.txt
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 --verify h8line:MMSA_H8C=0 h8c: fl8:MMSA_GEMM_FLAVOUR=8 > $O/ab.txt 2>&1
cat $O/ab.txt
for f in h8 h8c; do MMSA_ABLATE_FMT=$f timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/ablate_$f.txt 2>&1; done
cat $O/ablate_h8.txt $O/ablate_h8c.txt
