#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_ac; mkdir -p $O
( time timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5
( time timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4
python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['chains_per_gpu'], j['roofline']['frac'], j['cpu_baseline']['value'], j['verified'])"
