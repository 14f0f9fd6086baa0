"""Precision study on the CPU oracle (test infrastructure, never on the product path): what does it cost to hand the deformable-attention gather its `value`
tensor (value_proj output, ops/modules/ms_deform_attn.py:103-104) as fp16 -- the hi rows of h8c planes -- instead of fp32?  The gather is bound by the bytes it
pulls through the vector-memory pipe (2 KB of corner reads per query and head): fp16 values halve them.
Emulation: value_proj's output rounded to fp16 (round-to-nearest-even, clamped at +-57344 like the planes) in EVERY MSDeformAttn of the model; everything else
fp32.  Reports rel-L2 / max-rel of f1..f4 against the plain fp32 oracle.   python tools/msda_value_f16_study.py [tiny256|vitb512]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import ref_encoder as R
from tests.configs import CONFIGS, make_input
from tests.weights import seeded_state_dict

name = sys.argv[1] if len(sys.argv) > 1 else "vitb512"
cfg = CONFIGS[name]
torch.manual_seed(0)
torch.set_num_threads(8)
m = R.OracleEncoder(**cfg["kwargs"])
m.load_state_dict(seeded_state_dict(m, seed=cfg["seed"]))
x = make_input(cfg, batch=1)
with torch.no_grad():
    ref, _ = m(x)
n = 0
for mod in m.modules():
    if isinstance(mod, R.MSDeformAttn):
        lin = mod.value_proj
        lin.forward = (lambda t, lin=lin: torch.nn.functional.linear(t, lin.weight, lin.bias).clamp(-57344.0, 57344.0).half().float())
        n += 1
with torch.no_grad():
    got, _ = m(x)
print(f"{name}: value tensors of {n} MSDeformAttn modules rounded to fp16")
for i, (g, r) in enumerate(zip(got, ref)):
    rel = float((g.double() - r.double()).norm() / r.double().norm())
    mx = float((g - r).abs().max() / r.abs().max())
    print(f"  f{i+1}: rel-L2 {rel:.2e}  max-rel {mx:.2e}")
