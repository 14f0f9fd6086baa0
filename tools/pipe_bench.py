"""Experiment: two-stage pipelined encoder (SPM of batch i+1 under the ViT of batch i) vs the plain forward, graph replayed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
cfg = CONFIGS["vitl1024"]
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
x = make_input(cfg, batch=2, seed=1234).to(dev)
ref = [f.clone() for f in m(x)[0]]
torch.cuda.synchronize()


def graph_of(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    return g, out


def timeit(run, iters=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


PLAIN = os.environ.get("PB_PLAIN", "1") == "1"
if PLAIN:
    g_plain, _ = graph_of(lambda: m(x)[0])
assert m.forward_pipelined(x) is None          # prime: SPM of the first batch
for _ in range(2):
    outs = m.forward_pipelined(x)[0]
    if os.environ.get("PB_SYNC") == "1":
        torch.cuda.synchronize()
torch.cuda.synchronize()
print("eager pipelined == forward:", all(torch.equal(a, b) for a, b in zip(outs, ref)))
dbg = os.environ.get("PB_DBG", "")
if dbg == "noB":
    m._vit = lambda *a, **k: [torch.zeros(1, device=dev)]
if dbg == "noA":
    def _stub(x_, B_, H_, W_, c1_, c_, Nc_):
        c1_[:1].zero_()
        e = torch.cuda.Event(); e.record(torch.cuda.current_stream()); return e
    m._spm = _stub
m.multistream = os.environ.get("PB_MULTI", "1") == "1"
print("capturing, multistream =", m.multistream, flush=True)
gs = []
for par in range(2):   # two parities (buffer set 0 / 1): one call per graph, no warm-up call in between (it would flip the parity)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.forward_pipelined(x)[0]
    torch.cuda.synchronize()
    gs.append((g, out))
    print("captured parity", par, flush=True)
k = [0]


def run_pipe():
    gs[k[0] & 1][0].replay(); k[0] += 1


for rnd in range(2):
    if PLAIN:
        print(f"plain     {timeit(g_plain.replay):.2f} ms/step")
    print(f"pipelined {timeit(run_pipe):.2f} ms/step")
torch.cuda.synchronize()
print("graph pipelined == forward:", [all(torch.equal(a, b) for a, b in zip(g[1], ref)) for g in gs])
