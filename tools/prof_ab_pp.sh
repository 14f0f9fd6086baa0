cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pp in 0 1; do
  export MMSA_GEMM_PP=$pp
  rm -rf gpurun_out/prof_pp$pp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp$pp -- python bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-head > /dev/null 2>&1
  f=$(ls gpurun_out/prof_pp$pp/*/*kernel_stats.csv | head -1)
  echo PP=$pp; python tools/kstats.py $f 6 8
done
