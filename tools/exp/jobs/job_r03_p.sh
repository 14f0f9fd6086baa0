cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -v -m gpu --timeout 90 > $O/tests_ops.txt 2>&1; tail -5 $O/tests_ops.txt
timeout -k 10 600 python -m pytest tests/test_backbone_gpu.py -x -v -m gpu --timeout 200 --durations=8 > $O/tests_bb.txt 2>&1; tail -25 $O/tests_bb.txt
