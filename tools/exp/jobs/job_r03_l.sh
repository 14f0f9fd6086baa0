cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_l; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_planes_gpu.py tests/test_bookkeeping_gpu.py -x -q -m gpu -k "window or bookkeeping" 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
timeout -k 10 120 python tools/wattn_bench.py 1 > $O/wattn.txt 2>&1; timeout -k 10 120 python tools/wattn_bench.py 2 >> $O/wattn.txt 2>&1; grep wattn $O/wattn.txt
(bash tools/exp/slp_pkfma/build.sh && timeout -k 10 200 tools/exp/slp_pkfma/repro 200) > $O/slp.txt 2>&1; tail -3 $O/slp.txt
for S in 0 2002 2003 3002 2005; do echo "== stagger $S" >> $O/stagger.txt; V3_CHECK_FLAVOURS=8,8 MMSA_GEMM_STAGGER=$S timeout -k 10 200 python tools/v3_check.py time 0 2>&1 | grep "fl8 us" | head -2 >> $O/stagger.txt; done; cat $O/stagger.txt
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 --verify base: wattn3bar:MMSA_WATTN_P1=1 nomemset:MMSA_SKIP_MEMSET=1 cnx2:MMSA_H8=vit,inter,up,attnv,cnx2 > $O/ab.txt 2>&1; cat $O/ab.txt
