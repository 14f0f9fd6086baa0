"""Does the plain forward() stay bit-identical while an unrelated stream keeps the GPU busy?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
cfg = CONFIGS["vitl1024"]
torch.manual_seed(1234)
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
x = make_input(cfg, batch=2, seed=1234).to("cuda:0")
ref = [f.clone() for f in m(x)[0]]
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "matmul"
side = torch.cuda.Stream()
a = torch.randn(8192, 4096, device="cuda:0"); b = torch.randn(4096, 4096, device="cuda:0")
m2 = None
if mode == "model":
    m2 = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
    m2(x); torch.cuda.synchronize()
bad = 0
for it in range(int(os.environ.get("ITERS", "40"))):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        if mode == "matmul":
            for _ in range(40):
                c = a @ b
        elif mode == "model":
            m2(x)
    outs = m(x)[0]
    torch.cuda.synchronize()
    eq = [torch.equal(p, q) for p, q in zip(outs, ref)]
    if not all(eq):
        bad += 1
        print(it, eq, [f"{((p - q).abs().max() / q.abs().max()).item():.2e}" for p, q in zip(outs, ref)])
print(mode, "mismatching iterations:", bad)
