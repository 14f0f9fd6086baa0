cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ba; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_backbone_gpu.py -x -v -m gpu > $O/tests_bb.txt 2>&1; tail -6 $O/tests_bb.txt
timeout -k 10 700 python tools/ab_env.py --rounds 3 --steps 20 --verify fold: nofold:MMSA_FOLD_CNX_LN=0 > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt | cut -c1-230
