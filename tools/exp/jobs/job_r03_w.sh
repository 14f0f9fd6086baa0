cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_bookkeeping_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
AB_NO_HEAD=0 timeout -k 10 600 python tools/ab_step.py ab/lib_old.so ab/lib_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/lib_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
timeout -k 10 300 python bench.py --no-cpu-baseline 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; python -c "import json; d=json.load(open('$O/bench.json')); print(d['verified'])"
