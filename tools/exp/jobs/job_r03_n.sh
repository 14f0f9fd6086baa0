cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_n; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/dma_lanes.hip -o /tmp/dma_lanes 2>/dev/null && timeout -k 10 120 /tmp/dma_lanes > $O/dma_lanes.txt 2>&1; cat $O/dma_lanes.txt
timeout -k 10 600 python tools/ab_env.py --rounds 2 --steps 20 base: nomemset:MMSA_SKIP_MEMSET=1 > $O/ab_memset.txt 2>&1; cat $O/ab_memset.txt
