#!/bin/bash
# round 5, job q: full GPU suite + the default bench line on the current tree
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_q; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -n 5 $O/tests.txt
timeout -k 10 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_q/bench.json'))
for k in ('value','ms_per_step','encoder_only','chains_probe_ms','eager_plugin_api','config4_frame','vith1024','worst_case_precision'):
    print(k, json.dumps(d.get(k))[:260])
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','kernel_ms_per_step')})
print('verified', d['verified'])
PY
