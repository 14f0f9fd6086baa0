import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
import mmsa.inference as inf
from tests.configs import CONFIGS, HEAD_CONFIGS
from tests.weights import seeded_state_dict
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **CONFIGS["vitl1024"]["kwargs"]))
m.load_state_dict(seeded_state_dict(m, seed=CONFIGS["vitl1024"]["seed"]))
h = mmsa.build_head(dict(type="SegformerHead", **HEAD_CONFIGS["head_vitl"]["kwargs"])).to(dev)
h.load_state_dict(seeded_state_dict(h, seed=HEAD_CONFIGS["head_vitl"]["seed"]))
g = torch.Generator().manual_seed(7)
frame = torch.randn(1, 6, 1080, 1920, generator=g)
frame[:, 3:] = (torch.rand(1, 3, 1080, 1920, generator=g) < 0.05).float() * torch.rand(1, 3, 1080, 1920, generator=g)
frame = frame.to(dev)
logits = inf.slide_inference(m, h, frame, (1024, 1024), (640, 640), max_batch=6)
want = inf.argmax_map(logits)
want2 = inf.argmax_map(inf.slide_inference(m, h, frame, (1024, 1024), (640, 640), max_batch=6))
cm, unc = inf.slide_class_map(m, h, frame, (1024, 1024), (640, 640), max_batch=6)
torch.cuda.synchronize()
d = (cm != want)
print("canvas path twice equal:", torch.equal(want, want2), " uncovered", int(unc.item()), " differing pixels:", int(d.sum().item()), "of", d.numel())
if d.any():
    idx = d.nonzero()[:5]
    top2 = logits[0].topk(2, dim=0).values
    for i in idx:
        y, x = int(i[-2]), int(i[-1])
        print("pixel", y, x, "canvas class", int(want.reshape(1080, 1920)[y, x]), "fused class", int(cm.reshape(1080, 1920)[y, x]), "top-2 logits", top2[:, y, x].tolist())
