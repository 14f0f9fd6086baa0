import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
for name in (sys.argv[1:] or ["vitl1024"]):
    cfg = CONFIGS[name]
    torch.manual_seed(1234)
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    x = make_input(cfg, batch=2, seed=1234).to("cuda:0")
    from mmsa import lib as _lib
    keep, _lib.POISON_LDS = _lib.POISON_LDS, False
    m.multistream = os.environ.get("PT_MULTI", "1") == "1"
    ref = [f.clone() for f in m(x)[0]]
    torch.cuda.synchronize()
    _lib.POISON_LDS = keep
    m._ws.poison()
    torch.cuda.synchronize()
    outs = m(x)[0]
    torch.cuda.synchronize()
    print(name, "LDS poison on" if os.environ.get("MMSA_DEBUG_POISON_LDS") == "1" else "", "after workspace poison: equal", [torch.equal(a, b) for a, b in zip(outs, ref)], "finite", [bool(torch.isfinite(a).all()) for a in outs],
          [f"{((a - b).abs().max() / b.abs().max()).item():.2e}" for a, b in zip(outs, ref)])
