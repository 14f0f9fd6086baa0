"""Per-stage error attribution of the HIP path at ViT-L 1024^2 against the float64 oracle (tests/golden/model_vitl1024_f64.npz,
tools/oracle/make_f64.py): for every tap the rel-L2 / max-rel of the GPU tensor at the fixture's 4096 probe positions, for the
default operand formats and for bf16 hi/lo everywhere (h8_sites = ()).  GPU box:  python tools/error_budget.py [out.json]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))

from tests.configs import CONFIGS, make_input, probe_index  # noqa: E402
from tests.weights import seeded_state_dict  # noqa: E402
from tools.oracle.make_f64 import NPROBE, tap_seed  # noqa: E402

def tap_errors(taps, g):
    """{tap: (rel_l2, max_rel)} of GPU taps against the fixture's float64 probes (normalised by the FULL tensor's max-abs)."""
    out = {}
    for k, v in taps.items():
        if f"{k}_probe" not in g:
            continue
        ref = torch.from_numpy(g[f"{k}_probe"])
        assert tuple(g[f"{k}_shape"]) == tuple(v.shape), (k, tuple(v.shape), tuple(g[f"{k}_shape"]))
        pi = probe_index(v.numel(), NPROBE, seed=tap_seed(k))
        got = v.flatten()[pi.to(v.device)].double().cpu()
        d = got - ref
        out[k] = (float(d.norm() / ref.norm()), float(d.abs().max() / g[f"{k}_norm"][1]))
    return out


def run(name="vitl1024", sites=None):
    import mmsa
    cfg = CONFIGS[name]
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
    if sites is not None:
        m.h8_sites = tuple(sites)
    x = make_input(cfg).to("cuda:0")
    _, taps = m.forward_taps(x)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"model_{name}_f64.npz"))
    return tap_errors(taps, g), g


def main():
    res = {}
    for tag, sites in (("default", None), ("b3_everywhere", ())):
        errs, g = run(sites=sites)
        res[tag] = {k: list(v) for k, v in errs.items()}
    keys = [k for k in ("twin0 twin1 twin2 twin3 fuse0 fuse1 fuse2 fuse3 c1_map c_in x_in x0 c0 x1 c1 x2 c2 x3 c3 f1 f2 f3 f4".split()) if k in res["default"]]
    print(f"{'tap':7s} {'fp32 ref vs f64':>22s} {'GPU default vs f64':>24s} {'GPU b3 everywhere':>24s}")
    for k in keys:
        f32 = g[f"{k}_f32_err"]
        a, b = res["default"][k], res["b3_everywhere"][k]
        print(f"{k:7s} {f32[0]:10.2e} {f32[1]:10.2e}   {a[0]:10.2e} {a[1]:10.2e}   {b[0]:10.2e} {b[1]:10.2e}")
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump({"config": "vitl1024", "columns": ["rel_l2", "max_rel (of the tensor's max-abs)"], "errors_vs_float64": res,
                       "fp32_oracle_vs_float64": {k: [float(g[f"{k}_f32_err"][0]), float(g[f"{k}_f32_err"][1])] for k in keys}}, fh, indent=1)


if __name__ == "__main__":
    main()
