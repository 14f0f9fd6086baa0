cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in ab/lib_vf.so ab/lib_wgrid.so ab/lib_vf.so ab/lib_wgrid.so; do cp $lib multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so; echo $lib; python tools/wattn_bench.py 2 2>&1 | tail -1; python tools/wattn_bench.py 1 2>&1 | tail -1; done
cp ab/lib_wgrid.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
timeout 300 python -m pytest tests/test_planes_gpu.py tests/test_bookkeeping_gpu.py -x -q -m gpu -k "window or bookkeeping" 2>&1 | tail -2
timeout 300 python tools/ab_step.py ab/lib_vf.so ab/lib_wgrid.so
