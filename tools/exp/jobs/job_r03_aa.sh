cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_aa; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_msda_gpu.py tests/test_ops_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
bash tools/exp/jobs/job_r03_q.sh; grep -i "msda\|attn_kernel\|total" gpurun_out/r03_q/kstats.txt
python -c "import json; d=json.load(open('gpurun_out/r03_q/bench.json')); print(d['verified'])"
