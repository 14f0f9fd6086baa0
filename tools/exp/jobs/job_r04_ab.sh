#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_ab; mkdir -p $O
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(j['value'], j['config']['chains_per_gpu'], j['verified'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --chains 1 > $O/bench1.json 2> $O/bench1.err; tail -2 $O/bench1.err
python -c "
import json; j=json.loads(open('$O/bench1.json').read().strip().splitlines()[-1]); print(j['value'], j['config']['chains_per_gpu'], j['verified'])"
timeout -k 10 300 python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -2
