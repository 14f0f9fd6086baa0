cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ad; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
AB_NO_HEAD=0 timeout -k 10 600 python tools/ab_step.py ab/lib_old.so ab/lib_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/lib_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
bash tools/exp/jobs/job_r03_q.sh; grep -i "dwconv\|total" gpurun_out/r03_q/kstats.txt
python -c "import json; d=json.load(open('gpurun_out/r03_q/bench.json')); print(d['verified'])"
