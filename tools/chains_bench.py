"""Timing of mmsa.Chains (one model, tagged scratch buffers, per-call GEMM grid cap) against one chain, encoder only.
python tools/chains_bench.py [chains=2] [batch=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
from tests.weights import seeded_state_dict
cfg = CONFIGS["vitl1024"]
dev = torch.device("cuda:0")
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
x = make_input(cfg, batch=B, seed=1234).to(dev)
for _ in range(2):
    m(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m(x)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    m(x)
torch.cuda.synchronize()


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


t1 = timeit(g.replay)
ch = mmsa.Chains(m, None, n=nch).capture(x)
t2 = timeit(lambda: ch.replay(join=False))
t3 = timeit(lambda: ch.replay(join=True))
print(f"one chain x batch {B}: {t1:.2f} ms | {nch} chains free-running: {t2:.2f} ms | joined per pass: {t3:.2f} ms", flush=True)
# free-running with a forced phase offset: chain 1 starts `off` ms late (one spin kernel), 40 passes so that the tail amortises
for off in (0.0, 6.0, 12.0, 18.0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if off > 0:
        with torch.cuda.stream(ch.streams[1]):
            torch.cuda._sleep(int(off * 1e-3 * 100e6))     # s_memrealtime ticks at 100 MHz
    for _ in range(40):
        ch.replay(join=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"  offset {off:4.1f} ms: {dt / 40:.2f} ms per pass over 40 passes ({(dt - off) / 40:.2f} without the offset itself)", flush=True)
