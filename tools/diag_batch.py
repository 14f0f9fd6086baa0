"""Diagnostic: run-to-run determinism and batch invariance of the ViT-L backbone outputs (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
from tests.weights import seeded_state_dict
cfg = CONFIGS["vitl1024"]
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
x1 = make_input(cfg, batch=1).to("cuda:0")
def run(x):
    fs, _ = m(x)
    return [f.clone() for f in fs]
def cmp(a, b):
    return " ".join(f"{((p - q).abs().max() / q.abs().max()).item():.2e}" for p, q in zip(a, b))
ya = run(x1); yb = run(x1)
print("B=1 twice:", cmp(ya, yb))
for B in (2, 6):
    yB = run(x1.expand(B, -1, -1, -1).contiguous())
    for k in (0, B - 1):
        print(f"B={B} image {k} vs B=1:", cmp([f[k:k + 1] for f in yB], ya))
    yB2 = run(x1.expand(B, -1, -1, -1).contiguous())
    print(f"B={B} twice:", cmp(yB, yB2))
if len(sys.argv) > 1:
    m.multistream = False
    ya = run(x1); yb = run(x1)
    print("single-stream B=1 twice:", cmp(ya, yb))
    y6 = run(x1.expand(6, -1, -1, -1).contiguous())
    print("single-stream B=6 image 5 vs B=1:", cmp([f[5:6] for f in y6], ya))
