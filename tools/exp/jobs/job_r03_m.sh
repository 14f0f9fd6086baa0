cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_m; mkdir -p $O
L=multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
for V in nosplit split nosplit split; do cp ab/lib_$V.so $L; echo "== $V" >> $O/split_shapes.txt; V3_CHECK_FLAVOURS=8,8 timeout -k 10 200 python tools/v3_check.py time 0 2>&1 | grep -E "^ +lin1|h8 fl8 us" | head -2 >> $O/split_shapes.txt; done; cat $O/split_shapes.txt | cut -c1-160
AB_NO_HEAD=0 timeout -k 10 600 python tools/ab_step.py ab/lib_nosplit.so ab/lib_split.so > $O/ab_split.txt 2>&1; cat $O/ab_split.txt
cp ab/lib_split.so $L
(timeout -k 10 200 tools/exp/slp_pkfma/repro 200) > $O/slp.txt 2>&1; tail -3 $O/slp.txt
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 --verify base: nomemset:MMSA_SKIP_MEMSET=1 cnx2:MMSA_H8=vit,inter,up,attnv,cnx2 > $O/ab_env.txt 2>&1; cat $O/ab_env.txt
