cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_q; mkdir -p $O
timeout -k 10 300 python bench.py --no-cpu-baseline 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/prof
python tools/kstats.py $O/kernel_stats.csv 62 45 > $O/kstats.txt; grep -i "gram\|fill\|colstats\|total" $O/kstats.txt
