"""fp64 ground truth of the path at a named configuration, with per-stage taps.

Runs the CPU oracle (oracle/ref_encoder.py, pinned to the reference to 3e-7) twice on the configuration's seeded weights and
input -- once in float64, once in float32 -- and writes tests/golden/model_<name>_f64.npz:

  * per tap (c1, c_in, x_in, x0..x3, c0..c3, twin0..3, fuse0..3, f1..f4): 4096 probe values of the float64 run, the probe seed,
    the tensor's float64 L2 norm and max-abs, and the FULL-tensor error of the float32 oracle against float64 (rel-L2, max-rel):
    how much of a float32 implementation's distance to the reference is the reference's own rounding noise;
  * for f1..f4 also the float64 values at the probe positions of the reference golden (model_<name>.npz), so that
    reference-fp32-vs-fp64 can be stated from committed data.

Data only; needs neither /root/reference nor a GPU.  Usage: python tools/oracle/make_f64.py [vitl1024]
"""
import os
import sys
import time
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_encoder as R  # noqa: E402
from tests.configs import CONFIGS, make_input, probe_index  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402

NPROBE = 4096


def tap_seed(name):
    return 1000 + zlib.crc32(name.encode()) % 100000


def run(name, dtype, threads=8):
    cfg = CONFIGS[name]
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    orc = R.OracleEncoder(**cfg["kwargs"])
    orc.load_state_dict(seeded_state_dict(orc, seed=cfg["seed"]))
    orc = orc.to(dtype)
    x = make_input(cfg).to(dtype)
    taps = {}
    t0 = time.time()
    fs, _ = orc(x, taps)
    for i, f in enumerate(fs):
        taps[f"f{i + 1}"] = f
    print(f"{name} {dtype}: {time.time() - t0:.1f} s, {len(taps)} taps", flush=True)
    return {k: v.contiguous() for k, v in taps.items()}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vitl1024"
    t64 = run(name, torch.float64)
    t32 = run(name, torch.float32)
    out = {}
    rows = []
    for k in sorted(t64):
        a, b = t64[k], t32[k].double()
        nrm, mx = a.norm().item(), a.abs().max().item()
        d = b - a
        r, m = d.norm().item() / max(nrm, 1e-300), d.abs().max().item() / max(mx, 1e-300)
        pi = probe_index(a.numel(), NPROBE, seed=tap_seed(k))
        out[f"{k}_probe"] = a.flatten()[pi].numpy()
        out[f"{k}_shape"] = np.array(a.shape)
        out[f"{k}_norm"] = np.array([nrm, mx])
        out[f"{k}_f32_err"] = np.array([r, m])
        rows.append((k, tuple(a.shape), r, m))
    for i in range(4):   # the reference golden's own probe positions
        f = t64[f"f{i + 1}"]
        out[f"f{i + 1}_goldprobe"] = f.flatten()[probe_index(f.numel(), 2048, seed=100 + i)].numpy()
    path = os.path.join(ROOT, "tests", "golden", f"model_{name}_f64.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")
    print("tap: fp32 oracle vs fp64 oracle (rel-L2, max-rel)")
    for k, s, r, m in rows:
        print(f"  {k:8s} {str(s):24s} {r:.3e} {m:.3e}")
    gpath = os.path.join(ROOT, "tests", "golden", f"model_{name}.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        print("reference fp32 (golden probes) vs fp64 oracle at the same positions:")
        for i in range(4):
            ref = torch.from_numpy(g[f"f{i + 1}_probe"]).double()
            tru = torch.from_numpy(out[f"f{i + 1}_goldprobe"])
            print(f"  f{i + 1}: rel-L2 {((ref - tru).norm() / tru.norm()).item():.3e}  max-rel {((ref - tru).abs().max() / tru.abs().max()).item():.3e}")


if __name__ == "__main__":
    main()
