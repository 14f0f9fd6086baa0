"""Write the backbone's packed checkpoint once (mmsa.checkpoint.save_packed): plain state dict (Appendix A.3 keys) + the bf16 hi/lo
planes and folded tensors the HIP path consumes, so serving processes skip the re-split of 456 M parameters at their first forward.

  python tools/pack_checkpoint.py --config vitl1024 --pretrained sam_vit_l_image_encoder_no_neck.pth --convnext convnext_small.pth --out vitl.packed.pth
  python tools/pack_checkpoint.py --config vitl1024 --sam-release sam_vit_l_0b3195.pth --out vitl.packed.pth     (raw SAM release: converted like
                                                                                        segmentation/tools/SAM_checkpoint_convert.py:15-33)
Needs the GPU (the split kernels run once)."""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))


def main():
    import torch
    import mmsa
    from mmsa import checkpoint as C
    from tests.configs import CONFIGS
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vitl1024", choices=sorted(CONFIGS))
    ap.add_argument("--pretrained", default=None, help="converted SAM image-encoder checkpoint (init_weights path)")
    ap.add_argument("--sam-release", default=None, help="raw SAM release checkpoint: converted in memory first")
    ap.add_argument("--convnext", default=None, help="single-stream ConvNeXt checkpoint duplicated into both streams (TC:403-443)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    kw = dict(CONFIGS[a.config]["kwargs"])
    pre = a.pretrained
    if a.sam_release:
        conv = C.convert_sam_release(torch.load(a.sam_release, map_location="cpu"))
        fd, pre = tempfile.mkstemp(suffix=".pth")
        os.close(fd)
        torch.save(conv, pre)
    m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", pretrained=pre, **kw))
    if a.convnext:
        print("ConvNeXt keys loaded:", len(m.load_convnext_checkpoint(a.convnext)))
    if pre:
        loaded, skipped, unexpected = m._pretrained_report
        print(f"SAM keys loaded: {len(loaded)}, skipped (shape): {skipped}, unexpected: {len(unexpected)}")
    C.save_packed(m, a.out)
    print("wrote", a.out, f"{os.path.getsize(a.out) / 2**20:.1f} MiB")


if __name__ == "__main__":
    main()
