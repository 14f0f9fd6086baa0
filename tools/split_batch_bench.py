"""Experiment: the step's batch of 2 images as TWO independent batch-1 chains on two HIP streams (two graphs replayed concurrently)
against ONE batch-2 chain.  Rationale: with one persistent workgroup per CU every GEMM's tiles run in lockstep, so the epilogue
(stores at HBM speed, matrix pipe idle) and the k-loop (matrix pipe busy, memory idle) alternate chip-wide; two independent chains
de-synchronise.  MMSA_GEMM_MAX_GRID caps the persistent GEMM grids so that two kernels can be resident at once.
python tools/split_batch_bench.py [batch_per_chain=1] [chains=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
from tests.weights import seeded_state_dict
cfg = CONFIGS["vitl1024"]
dev = torch.device("cuda:0")
bpc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sd = None
models, xs, graphs, streams = [], [], [], []
for r in range(nch):
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    if sd is None:
        sd = seeded_state_dict(m, seed=cfg["seed"])
    m.load_state_dict(sd)
    x = make_input(cfg, batch=bpc, seed=1234 + r).to(dev)
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        m(x)
    torch.cuda.synchronize()
    models.append(m); xs.append(x); graphs.append(g); streams.append(torch.cuda.Stream())


def run(n_rep, iters=10):
    for _ in range(2):
        for r in range(n_rep):
            with torch.cuda.stream(streams[r]):
                graphs[r].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for r in range(n_rep):
            with torch.cuda.stream(streams[r]):
                graphs[r].replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return dt * 1e3, bpc * n_rep / dt


for rnd in range(2):
    for n in sorted({1, nch}):
        ms, ips = run(n)
        print(f"grid cap {os.environ.get('MMSA_GEMM_MAX_GRID', 'none')}: {n} chain(s) x batch {bpc}: {ms:.2f} ms per round, {ips:.2f} images/s", flush=True)
