#!/bin/bash
# round 5, job g: new host features -- guard handles, mixed-state golden, cffn-512 fix, then the full bench line with its three new legs
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_inference_gpu.py tests/test_backbone_gpu.py -m gpu -x -q -k "chains or slide_runner or mixed or cffn or peaky or constructor" > $O/tests.txt 2>&1; tail -n 6 $O/tests.txt
timeout -k 10 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_g/bench.json'))
for k in ('value','ms_per_step','encoder_only','chains_probe_ms','eager_plugin_api','config4_frame','vith1024','worst_case_precision'):
    print(k, json.dumps(d.get(k))[:600])
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','kernel_ms_per_step')})
print('verified', d['verified'])
PY
