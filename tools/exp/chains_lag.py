"""Experiment (round 5): two free-running chains with an initial LAG between them, so that one chain's GEMM-heavy ViT phase runs beside the other's memory-bound spatial-prior phase
instead of both being in the same phase.  Steady-state ms per pass (2 images) for several lags.    python tools/exp/chains_lag.py [passes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch  # noqa: E402
import mmsa  # noqa: E402
from tests.configs import CONFIGS, HEAD_CONFIGS, make_input  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **CONFIGS["vitl1024"]["kwargs"]))
m.load_state_dict(seeded_state_dict(m, seed=CONFIGS["vitl1024"]["seed"]))
h = mmsa.build_head(dict(type="SegformerHead", **HEAD_CONFIGS["head_vitl"]["kwargs"])).to(dev)
h.load_state_dict(seeded_state_dict(h, seed=HEAD_CONFIGS["head_vitl"]["seed"]))
x = make_input(CONFIGS["vitl1024"], batch=2).to(dev)
ch = mmsa.Chains(m, h, n=2, check_every=1000).capture(x)
for _ in range(3):
    ch.replay().unverified
torch.cuda.synchronize()
clock_khz = torch.cuda.get_device_properties(dev).clock_rate if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2400000


def run(lag_ms):
    torch.cuda.synchronize()
    evs = [[], []]
    if lag_ms > 0:
        with torch.cuda.stream(ch.streams[1]):
            torch.cuda._sleep(int(lag_ms * 1e-3 * 2.4e9))     # a one-thread spin kernel: delays the stream, occupies nothing
    t0 = time.perf_counter()
    for k in range(K):
        for i, (g, s) in enumerate(zip(ch.graphs, ch.streams)):
            with torch.cuda.stream(s):
                g.replay()
                e = torch.cuda.Event(enable_timing=True)
                e.record(s)
                evs[i].append(e)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    # steady state: chain i's passes 4 .. K-1
    per = [evs[i][4].elapsed_time(evs[i][K - 1]) / (K - 1 - 4) for i in range(2)]
    lag_end = evs[0][K - 1].elapsed_time(evs[1][K - 1])
    return per, wall / K, lag_end


for lag in (0, 0, 6, 10, 14, 18, 22, 0):
    per, wall, lag_end = run(lag)
    print(f"initial lag {lag:3d} ms: steady-state period chain0 {per[0]:.3f} ms, chain1 {per[1]:.3f} ms per pass of 2 images (= {2e3 / max(per):.2f} images/s); "
          f"wall / pass incl. ramp {wall:.3f} ms; chain1 finishes {lag_end:.2f} ms after chain0", flush=True)
