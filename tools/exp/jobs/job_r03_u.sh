cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_u; mkdir -p $O
for CH in 2 1; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof$CH -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-verify --chains $CH --force-chains > $O/bench$CH.txt 2>&1
T=$(ls $O/prof$CH/*/*kernel_trace.csv | head -1)
echo "== chains $CH" >> $O/gaps.txt; tail -1 $O/bench$CH.txt | cut -c1-160 >> $O/gaps.txt
python tools/timeline_gaps.py $T 20 >> $O/gaps.txt 2>&1
rm -rf $O/prof$CH
done
cat $O/gaps.txt
