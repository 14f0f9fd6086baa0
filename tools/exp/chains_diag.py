"""Why do two chains serialise?  One variant per process: python tools/exp/chains_diag.py  (env: DIAG_GUARD=off|sync, DIAG_MS=0|1, DIAG_CAP=auto|<n>,
DIAG_ROOT=<repo root to import from>)"""
import os, sys, time
ROOT = os.environ.get("DIAG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
from tests.weights import seeded_state_dict
cfg = CONFIGS["vitl1024"]
dev = torch.device("cuda:0")
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
if os.environ.get("DIAG_GUARD") and hasattr(m, "attention_guard"):
    m.attention_guard = os.environ["DIAG_GUARD"]
if os.environ.get("DIAG_MS") is not None and os.environ.get("DIAG_MS") != "":
    m.multistream = os.environ["DIAG_MS"] != "0"
x = make_input(cfg, batch=2, seed=1234).to(dev)
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
cap = os.environ.get("DIAG_CAP", "auto")
if cap != "auto":
    # capture with a fixed cap: patch the device CU count the Chains class divides
    import types
    real = torch.cuda.get_device_properties
    class P:  # noqa: E701
        def __init__(self, p): self.p = p
        def __getattr__(self, k): return 2 * int(cap) if k == "multi_processor_count" else getattr(self.p, k)
    torch.cuda.get_device_properties = lambda d=None: P(real(d))
ch = mmsa.Chains(m, None, n=2).capture(x)
t2 = timeit(lambda: ch.replay(join=True))
# one chain's graph alone
def one():
    with torch.cuda.stream(ch.streams[0]):
        ch.graphs[0].replay()
t3 = timeit(one)
print(f"[{os.environ.get('DIAG_TAG', '')}] one chain x 2 images {t1:.2f} ms | 2 chains joined {t2:.2f} ms | chain 0 alone (1 image, capped grid) {t3:.2f} ms", flush=True)
