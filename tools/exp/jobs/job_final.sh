cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_z
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_z/smoke.txt 2>&1; tail -1 gpurun_out/r03_z/smoke.txt
bash tools/round_end.sh r03_z && bash tools/pmc_all.sh r03
