"""Experiment: how much throughput does running two independent encoder steps concurrently (two graphs on two streams) add on
one GPU?  An upper bound for pipelining the SPM of batch i+1 under the ViT blocks of batch i."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
cfg = CONFIGS["vitl1024"]
dev = torch.device("cuda:0")
models, xs, graphs, streams = [], [], [], []
for r in range(2):
    torch.manual_seed(1234)
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    x = make_input(cfg, batch=2, seed=1234 + r).to(dev)
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
    return dt * 1e3, 2 * n_rep / dt


for rnd in range(2):
    for n in (1, 2):
        ms, ips = run(n)
        print(f"{n} replica(s): {ms:.2f} ms per round, {ips:.2f} images/s", flush=True)
