"""Generate golden vectors by running the UNMODIFIED reference (imported through
tools/oracle/ref_import.py) in the build container.  Output: tests/golden/*.npz (data only:
inputs, seeds, expected outputs).  Re-run:  python tools/oracle/make_golden.py [--big]

Weights are not stored: they come from the build's deterministic generator
`tests.weights.seeded_state_dict(model, seed)` applied to the reference model with
load_state_dict; a checksum of the generated weights is stored so drift is detected.
"""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import ref_import  # noqa: E402
from tests.weights import large_magnitude, peaky_attention, seeded_state_dict  # noqa: E402
from tests.configs import CONFIGS, HEAD_CONFIGS, make_head_inputs, make_input, weights_checksum, probe_index  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_msda():
    fn = ref_import.ref_functions()["msda_core"]
    out = {}
    # the reference's own known-answer fixture, OPS/test.py:16-33
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    out.update(t_shapes=shapes.numpy(), t_lsi=lsi.numpy(), t_value=value.numpy(), t_loc=loc.numpy(), t_aw=aw.numpy(),
               t_out=fn(value, shapes, loc, aw).numpy(),
               t_out64=fn(value.double(), shapes, loc.double(), aw.double()).numpy())
    # injector-like (3 levels) and extractor-like (1 level) cases with out-of-range samples
    g = torch.Generator().manual_seed(11)
    for tag, shp, Lq in (("inj", [(16, 12), (8, 6), (4, 3)], 48), ("ext", [(8, 6)], 252)):
        shapes = torch.as_tensor(shp, dtype=torch.long)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        N, M, D, L, P = 2, 4, 32, len(shp), 4
        value = torch.randn(N, S, M, D, generator=g)
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15  # ~23% of taps outside [0,1]
        aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
        out.update({f"{tag}_shapes": shapes.numpy(), f"{tag}_lsi": lsi.numpy(), f"{tag}_value": value.numpy(),
                    f"{tag}_loc": loc.numpy(), f"{tag}_aw": aw.numpy(), f"{tag}_out": fn(value, shapes, loc, aw).numpy()})
    np.savez_compressed(os.path.join(OUT, "msda.npz"), **out)


def gen_msda_bwd():
    """Gradients of the reference's own ms_deform_attn_core_pytorch (OPS/functions/ms_deform_attn_func.py:53-75), taken by autograd
    for a seeded upstream gradient: what ms_deform_attn_backward (OPS/src/vision.cpp:15) must return.  Cases: the geometry of the
    reference's gradient check (OPS/test.py:16-20,77-90: N,M = 1,2; Lq,L,P = 2,2,2; shapes (6,4),(3,2); channels 4) in float64,
    and the border-sampling injector / extractor-like cases of msda.npz (23 % of taps outside [0,1]) plus a D = 40 head (ViT-H) in
    float32."""
    fn = ref_import.ref_functions()["msda_core"]
    out = {}

    def case(tag, shp, N, M, D, Lq, P, dt, seed, spread):
        g = torch.Generator().manual_seed(seed)
        shapes = torch.as_tensor(shp, dtype=torch.long)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S, L = int(shapes.prod(1).sum()), len(shp)
        value = (torch.randn(N, S, M, D, generator=g, dtype=torch.float64)).to(dt).requires_grad_(True)
        loc = (torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * spread - (spread - 1) / 2).to(dt).requires_grad_(True)
        aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g, dtype=torch.float64), -1).view(N, Lq, M, L, P).to(dt).requires_grad_(True)
        gout = torch.randn(N, Lq, M * D, generator=g, dtype=torch.float64).to(dt)
        y = fn(value, shapes, loc, aw)
        gv, gl, ga = torch.autograd.grad(y, (value, loc, aw), gout)
        out.update({f"{tag}_shapes": shapes.numpy(), f"{tag}_lsi": lsi.numpy(), f"{tag}_value": value.detach().numpy(),
                    f"{tag}_loc": loc.detach().numpy(), f"{tag}_aw": aw.detach().numpy(), f"{tag}_gout": gout.numpy(),
                    f"{tag}_gvalue": gv.numpy(), f"{tag}_gloc": gl.numpy(), f"{tag}_gaw": ga.numpy()})

    case("t64", [(6, 4), (3, 2)], 1, 2, 4, 2, 2, torch.float64, 3, 1.0)
    case("inj", [(16, 12), (8, 6), (4, 3)], 2, 4, 32, 48, 4, torch.float32, 21, 1.3)
    case("ext", [(8, 6)], 2, 4, 32, 252, 4, torch.float32, 22, 1.3)
    case("d40", [(8, 6), (4, 3)], 1, 2, 40, 24, 4, torch.float32, 23, 1.3)
    np.savez_compressed(os.path.join(OUT, "msda_bwd.npz"), **out)
    print("msda_bwd", sorted(k for k in out if k.endswith("_gvalue")), flush=True)


def gen_bookkeeping():
    f = ref_import.ref_functions()
    out = {}
    # window partition/unpartition on an index tensor: bit-exact token bookkeeping (IE:504-551)
    for (H, W, ws) in ((14, 14, 14), (16, 16, 14), (20, 20, 14), (64, 64, 14), (32, 32, 14)):
        idx = torch.arange(1, 2 * H * W + 1, dtype=torch.float32).view(2, H, W, 1)
        win, pad_hw = f["window_partition"](idx, ws)
        back = f["window_unpartition"](win, ws, pad_hw, (H, W))
        out[f"wp_{H}_{W}_{ws}"] = win.squeeze(-1).to(torch.int64).numpy()
        out[f"wu_{H}_{W}_{ws}"] = back.squeeze(-1).to(torch.int64).numpy()
    # get_rel_pos gather table incl. linear interpolation branch (IE:554-584)
    g = torch.Generator().manual_seed(7)
    for (q, L) in ((14, 27), (64, 127), (14, 31), (20, 31), (16, 31), (32, 127)):
        table = torch.randn(L, 8, generator=g)
        out[f"rp_in_{q}_{L}"] = table.numpy()
        out[f"rp_out_{q}_{L}"] = f["get_rel_pos"](q, q, table).numpy()
    np.savez_compressed(os.path.join(OUT, "bookkeeping.npz"), **out)


def gen_model(name, full):
    cfg = CONFIGS[name]
    ref = ref_import.build_reference(_withcp=cfg.get("type", "").endswith("withcp"), **cfg["kwargs"])
    keys = list(ref.state_dict().keys())
    shapes = [list(v.shape) for v in ref.state_dict().values()]
    sd = seeded_state_dict(ref, seed=cfg["seed"])
    if cfg.get("qk_scale"):
        sd = peaky_attention(sd, cfg["kwargs"]["embed_dim"], cfg["qk_scale"], cfg.get("qk_blocks"))
    if cfg.get("large_mag"):
        sd = large_magnitude(sd, cfg["large_mag"])
    ref.load_state_dict(sd)
    x = make_input(cfg)
    with torch.no_grad():
        fs, none = ref(x)
    assert none is None
    out = dict(weights_checksum=np.float64(weights_checksum(sd)), x_checksum=np.float64(x.double().abs().sum().item()))
    for i, f in enumerate(fs):
        f = f.contiguous()
        out[f"f{i+1}_shape"] = np.array(f.shape)
        out[f"f{i+1}_stats"] = np.array([f.double().mean().item(), f.double().abs().mean().item(), f.abs().max().item(),
                                         f.double().pow(2).sum().sqrt().item()])
        pi = probe_index(f.numel(), 2048, seed=100 + i)
        out[f"f{i+1}_probe"] = f.flatten()[pi].numpy()
        if full:
            out[f"f{i+1}"] = f.numpy() if i > 0 or f.shape[-1] <= 56 else f[..., ::2, ::2].contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **out)
    if name in ("tiny256_plain", "tiny256_norel"):
        with open(os.path.join(OUT, f"state_dict_keys_{name.replace('256', '')}.txt"), "w") as fh:
            for k, s in zip(keys, shapes):
                fh.write(f"{k} {s}\n")
    if name == "tiny224":
        with open(os.path.join(OUT, "state_dict_keys_tiny.txt"), "w") as fh:
            for k, s in zip(keys, shapes):
                fh.write(f"{k} {s}\n")
    if name == "vitl1024":
        with open(os.path.join(OUT, "state_dict_keys_vitl.txt"), "w") as fh:
            for k, s in zip(keys, shapes):
                fh.write(f"{k} {s}\n")
    print(name, [tuple(f.shape) for f in fs], flush=True)


def gen_head(name, full):
    """Reference SegformerHead (imported unmodified; see ref_import.build_reference_head) on seeded weights + inputs."""
    cfg = HEAD_CONFIGS[name]
    ref = ref_import.build_reference_head(**cfg["kwargs"])
    sd = seeded_state_dict(ref, seed=cfg["seed"])
    ref.load_state_dict(sd)
    xs = make_head_inputs(cfg)
    with torch.no_grad():
        y = ref(xs).contiguous()
    out = dict(weights_checksum=np.float64(weights_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})),
               shape=np.array(y.shape),
               stats=np.array([y.double().mean().item(), y.double().abs().mean().item(), y.abs().max().item()]),
               probe=y.flatten()[probe_index(y.numel(), 4096, seed=200)].numpy())
    if full:
        out["logits"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    with open(os.path.join(OUT, "state_dict_keys_head.txt"), "w") as fh:
        for k, v in ref.state_dict().items():
            fh.write(f"{k} {list(v.shape)}\n")
    print(name, tuple(y.shape), flush=True)


def gen_slide():
    """The reference's own EncoderDecoder.slide_inference on a seeded frame with a fixed toy encode_decode (tests/golden/slide.npz)."""
    from tests.configs import toy_encode_decode
    out = {}
    for tag, (hw, crop, stride) in dict(a=((90, 150), (64, 64), (40, 40)), b=((64, 100), (64, 64), (48, 48)), c=((70, 70), (64, 64), (64, 64))).items():
        g = torch.Generator().manual_seed(31)
        img = torch.randn(2, 6, hw[0], hw[1], generator=g)
        fn = toy_encode_decode(5, seed=77)
        y = ref_import.reference_slide_inference(fn, img, crop, stride, 5)
        out[f"{tag}_cfg"] = np.array(list(hw) + list(crop) + list(stride))
        out[f"{tag}_out"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "slide.npz"), **out)
    print("slide", {k: v.shape for k, v in out.items() if k.endswith("_out")}, flush=True)


def gen_whole_dim():
    """The reference's own EncoderDecoder.whole_inference_dim / whole_inference_dim_cut (the test modes of the DELIVER and FMB configs:
    test_cfg mode='whole_dim' dim=(1024,1024); mode='whole_dim_cut' dim=(600,800) cut_dim=(800,600)) on a seeded image with the fixed toy
    encode_decode of gen_slide (tests/golden/whole_dim.npz)."""
    from tests.configs import toy_encode_decode
    out = {}
    cases = dict(a=((64, 64), (64, 64), None, True),          # 'whole_dim', dim == input size (DELIVER)
                 b=((64, 80), (48, 60), None, True),          # 'whole_dim' to another size
                 c=((80, 80), (60, 80), (80, 60), False),     # 'whole_dim_cut' as the FMB configs run it: no rescale, crop to [.., :60, :80]
                 d=((80, 80), (60, 80), (70, 50), True))      # 'whole_dim_cut' with rescale
    for tag, (hw, dim, cut, rescale) in cases.items():
        g = torch.Generator().manual_seed(33)
        img = torch.randn(2, 6, hw[0], hw[1], generator=g)
        fn = toy_encode_decode(5, seed=78)
        y = ref_import.reference_whole_dim(fn, img, dim, cut, rescale)
        assert isinstance(y, tuple) and y[1] == (None,)
        out[f"{tag}_cfg"] = np.array(list(hw) + list(dim) + (list(cut) if cut else [0, 0]) + [int(rescale)])
        out[f"{tag}_out"] = y[0].numpy()
    # rescale=False in 'whole_dim' mode: the reference's method falls off its end (ED:334-346) and returns None
    assert ref_import.reference_whole_dim(toy_encode_decode(5, seed=78), torch.zeros(1, 6, 64, 64), (64, 64), None, False) is None
    np.savez_compressed(os.path.join(OUT, "whole_dim.npz"), **out)
    print("whole_dim", {k: v.shape for k, v in out.items() if k.endswith("_out")}, flush=True)


def gen_ckpt():
    """TwinConvNeXt.init_weights (TC:403-443) of the reference on a seeded single-stream ConvNeXt checkpoint: which twin keys
    end up loaded, and their checksums (tests/golden/convnext_ckpt.npz)."""
    import tempfile
    from tests.configs import fake_convnext_checkpoint
    cfg = CONFIGS["tiny256"]
    ref0 = ref_import.build_reference(**cfg["kwargs"])
    twin = [(k[len("spm.twin_conv."):], tuple(v.shape)) for k, v in ref0.state_dict().items() if k.startswith("spm.twin_conv.")]
    ck = fake_convnext_checkpoint(twin)
    path = os.path.join(tempfile.gettempdir(), "mmsa_fake_convnext.pth")
    torch.save({"state_dict": ck}, path)
    torch.manual_seed(0)
    ref = ref_import.build_reference(**dict(cfg["kwargs"], checkpoint=path))
    sd = ref.state_dict()
    keys = [k for k in sd if k.startswith("spm.twin_conv.")]
    # a key counts as loaded when its tensor equals the checkpoint tensor it would be fed from
    loaded = []
    for k in keys:
        sub = k[len("spm.twin_conv."):]
        first, rest = sub.split(".", 1)
        src = (first[:-2] if first.endswith(("_x", "_y")) else first) + "." + rest
        if src in ck and ck[src].shape == sd[k].shape and torch.equal(ck[src], sd[k]):
            loaded.append(k)
    np.savez_compressed(os.path.join(OUT, "convnext_ckpt.npz"), loaded=np.array(loaded), n_twin=np.int64(len(keys)),
                        checksum=np.float64(sum(sd[k].double().abs().sum().item() for k in loaded)))
    print("convnext ckpt: loaded", len(loaded), "of", len(keys), "twin keys", flush=True)


def gen_sam_ckpt():
    """The reference's OWN checkpoint ingestion on seeded SAM-style checkpoints (tests/golden/sam_ckpt.npz):
      A. `mmcv_custom.load_checkpoint` (checkpoint.py:319-514) as called by ImageEncoderViT.init_weights (IE:305-315) on a wrapped
         checkpoint: {'state_dict': {'module.<key>': ...}} with an unexpected key, a missing block and one rel-pos table whose
         length does not match the model's (skipped with a warning by the non-strict loader);
      B. a raw SAM release layout ('image_encoder.<key>' + 'image_encoder.neck.*' + prompt/mask-decoder keys) passed through the
         reference's tools/SAM_checkpoint_convert.py::remove_neck_from_checkpoint (:15-33) and then loaded the same way.
    Stored: which backbone keys end up equal to their checkpoint tensor, and a checksum of the whole loaded ViT part."""
    import importlib.util
    import tempfile
    from tests.configs import fake_sam_checkpoint
    cfg = CONFIGS["tiny256"]
    ref0 = ref_import.build_reference(**cfg["kwargs"])
    import mmcv_custom.checkpoint as ck          # the reference's loader; its two mmcv helpers are absent from this image
    ck.is_module_wrapper = lambda m: False       # mmcv.parallel.is_module_wrapper: no DataParallel wrapper here
    ck.get_dist_info = lambda: (0, 1)            # mmcv.runner.get_dist_info: single process
    vit = [(k, tuple(v.shape)) for k, v in ref0.state_dict().items()
           if k.startswith(("pos_embed", "patch_embed.", "blocks."))]
    tmpd = tempfile.gettempdir()
    out = {}

    def run(tag, path, plain):
        torch.manual_seed(0)
        ref = ref_import.build_reference(**dict(cfg["kwargs"], pretrained=path))
        sd = ref.state_dict()
        loaded = [k for k, _ in vit if k in plain and plain[k].shape == sd[k].shape and torch.equal(plain[k], sd[k])]
        out[f"{tag}_loaded"] = np.array(loaded)
        out[f"{tag}_not_loaded"] = np.array([k for k, _ in vit if k not in loaded])
        out[f"{tag}_checksum"] = np.float64(sum(sd[k].double().abs().sum().item() for k in loaded))   # the keys left alone keep an initialisation that is the model's own
        print("sam ckpt", tag, "loaded", len(loaded), "of", len(vit), flush=True)

    # A: wrapped + prefixed
    plain = fake_sam_checkpoint(vit, seed=51)
    pa = os.path.join(tmpd, "mmsa_fake_sam_a.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in plain.items()}, "meta": {"epoch": 3}}, pa)
    run("a", pa, plain)
    # B: raw SAM layout through the reference's converter
    raw = {"image_encoder." + k: v for k, v in fake_sam_checkpoint(vit, seed=52, drop=False).items()}
    raw["image_encoder.neck.0.weight"] = torch.ones(4, 4)
    raw["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = torch.ones(2, 8)
    raw["mask_decoder.iou_token.weight"] = torch.ones(1, 8)
    praw, pconv = os.path.join(tmpd, "mmsa_fake_sam_raw.pth"), os.path.join(tmpd, "mmsa_fake_sam_conv.pth")
    torch.save(raw, praw)
    spec = importlib.util.spec_from_file_location("sam_convert", os.path.join(ref_import.SEG, "tools", "SAM_checkpoint_convert.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        conv.remove_neck_from_checkpoint(praw, pconv)
    converted = torch.load(pconv, map_location="cpu")
    out["b_converted_keys"] = np.array(sorted(converted.keys()))
    run("b", pconv, converted)
    np.savez_compressed(os.path.join(OUT, "sam_ckpt.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also ViT-B@512 and ViT-L@1024 (minutes, GBs of RAM)")
    ap.add_argument("--only", default=None, help="run one generator: msda_bwd | sam_ckpt | model:<config name>")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if a.only:
        if a.only.startswith("model:"):
            n = a.only.split(":", 1)[1]
            gen_model(n, full=n.startswith("tiny"))
        else:
            globals()["gen_" + a.only]()
        return
    gen_msda_bwd()
    gen_msda()
    gen_bookkeeping()
    for n in ("tiny224", "tiny256", "tiny320", "tiny256_plain", "tiny256_norel", "tiny256_wide"):
        gen_model(n, full=True)
    gen_slide()
    gen_whole_dim()
    gen_ckpt()
    gen_head("head_vitl", full=False)
    gen_head("head_odd", full=True)
    gen_head("head_tiny", full=True)
    if a.big:
        gen_model("vitb512", full=False)
        gen_model("vitl1024", full=False)
        gen_model("vitl1024_b", full=False)
        gen_model("vitl1024_peaky", full=False)
        gen_model("vitl1024_mixed", full=False)
        gen_model("vitl1024_wide", full=False)


if __name__ == "__main__":
    main()
