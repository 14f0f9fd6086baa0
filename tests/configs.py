"""Model configurations and synthetic-input recipes shared by the golden generator
(tools/oracle/make_golden.py), the tests and bench.py.  Data only; no reference code."""
import torch

_TINY = dict(modalities_name=["rgb", "lidar"], modalities_ch=[3, 3], init_values=1e-6, patch_size=16,
             embed_dim=64, depth=4, num_heads=2, mlp_ratio=4, drop_path_rate=0.3, drop_multimodal_path=0,
             conv_inplane=16, n_points=4, deform_num_heads=2, cffn_ratio=0.25, deform_ratio=0.5, with_cp=True,
             interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]], global_attn_indexes=[1, 3], window_size=14,
             arch={"depths": [1, 1, 1, 1], "channels": [32, 64, 128, 256]}, pretrained_size=256)

# CFG-L = segmentation/configs/DELIVER/Segformer_MMSAM_adapter_large_DELIVER_1024x1024_ss_RGBLIDAR.py:30-56
_VITL = dict(img_size=1024, modalities_name=["rgb", "lidar"], modalities_ch=[3, 3], init_values=1e-6,
             gamma_init_values=1e-6, patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
             drop_path_rate=0.3, drop_multimodal_path=0, conv_inplane=48, n_points=4, deform_num_heads=16,
             cffn_ratio=0.25, deform_ratio=0.5, with_cp=True,
             interaction_indexes=[[0, 5], [6, 11], [12, 17], [18, 23]], global_attn_indexes=[5, 11, 17, 23],
             window_size=14, arch="small")

# BASELINE.json configs[0]: ViT-B SAM encoder + adapter, 512x512
_VITB = dict(_VITL, img_size=512, embed_dim=768, depth=12, num_heads=12, deform_num_heads=12,
             interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]], global_attn_indexes=[2, 5, 8, 11])

# the pinnable half of BASELINE.json configs[4]: SAM ViT-H (IE:188-303 with embed_dim 1280, depth 32, 16 heads -> head_dim 80,
# global blocks per the comment at IE:206) behind the same RGB+LiDAR adapter.  NOT the headline workload; the 3-modality / fp8 part
# of that config has no reference implementation (TC:296-316).
_VITH = dict(_VITL, embed_dim=1280, depth=32, num_heads=16, deform_num_heads=16,
             interaction_indexes=[[0, 7], [8, 15], [16, 23], [24, 31]], global_attn_indexes=[7, 15, 23, 31])

# FMB family (configs/FMB/Segformer_MMSAM_adapter_large_FMB_800x800_ss_RGBTHERM.py:14,26-49): the ...NEWwithcp class at img_size 800 --
# a 50 x 50 token grid: window padding 50 -> 56, pos-embed bicubic 64 -> 50, global rel-pos tables 127 -> 99 rows by linear
# interpolation, GFFM LayerNorm length 200^2; at ViT-L head_dim 64 the global blocks take the attention kernel with a rel-pos prepass
_VITL800 = dict(_VITL, img_size=800, conv_drop_path_rate=0.4)

CONFIGS = {
    "tiny224": dict(kwargs=dict(_TINY, img_size=224), batch=1, seed=1, in_seed=5),
    "tiny256": dict(kwargs=dict(_TINY, img_size=256), batch=1, seed=2, in_seed=6),
    "tiny320": dict(kwargs=dict(_TINY, img_size=320), batch=1, seed=3, in_seed=7),
    # the constructor switches no shipped config turns off (BK:32-34): extractors without ConvFFN (AM:485-488), no extra extractors in
    # the last interaction (BK:91-92), no ViT feature added in the tail (BK:326)
    # a ViT without relative position tables and without a qkv bias (IE:317,320-327): the kernels run on zero stand-ins
    "tiny256_norel": dict(kwargs=dict(_TINY, img_size=256, use_rel_pos=False, qkv_bias=False), batch=1, seed=23, in_seed=24),
    "tiny256_plain": dict(kwargs=dict(_TINY, img_size=256, with_cffn=False, use_extra_extractor=False, add_vit_feature=False),
                          batch=1, seed=21, in_seed=22),
    "vitb512": dict(kwargs=_VITB, batch=1, seed=4, in_seed=8),
    "vitl1024": dict(kwargs=_VITL, batch=1, seed=5, in_seed=9),
    # the second image of the benchmarked batch (image 1 was covered by graph == eager and batch invariance only): same weights,
    # another seeded input, its own reference probes
    "vitl1024_b": dict(kwargs=_VITL, batch=1, seed=5, in_seed=39),
    # peaky attention at ViT-L: the seeded weights with the q / k rows of every qkv projection x 3 (tests/weights.py
    # peaky_attention): max |logit| ~ 30-40, where fp16 attention operands leave the gate and the blocks must run bf16 hi/lo
    "vitl1024_peaky": dict(kwargs=_VITL, batch=1, seed=5, in_seed=9, qk_scale=3.0),
    # a MIXED precision state at ViT-L: q / k x 3 in 6 of the 24 blocks only (two of them global blocks) -- the guard moves those six to
    # fp16 hi/lo pairs, the other eighteen stay on single fp16 operands, in one forward
    "vitl1024_mixed": dict(kwargs=_VITL, batch=1, seed=5, in_seed=9, qk_scale=3.0, qk_blocks=[2, 5, 9, 14, 19, 23]),
    # values beyond the fp16-based operand formats' range inside the ViT blocks (tests/weights.py large_magnitude): a post-LayerNorm channel of
    # ~1e5 in front of qkv and of lin1, a GELU hidden unit of 6e4 -- the reference is fp32 and computes them; the model must go to its wide-range state and agree
    "tiny256_wide": dict(kwargs=dict(_TINY, img_size=256), batch=1, seed=2, in_seed=6, large_mag=dict(ln2=(1, 5), ln1=(2, 9), gelu=(3, 17))),
    "vitl1024_wide": dict(kwargs=_VITL, batch=1, seed=5, in_seed=9, large_mag=dict(ln2=(7, 100), ln1=(12, 333), gelu=(20, 1234))),
    "vith1024": dict(kwargs=_VITH, batch=1, seed=17, in_seed=18),
    "vitl800": dict(kwargs=_VITL800, batch=1, seed=19, in_seed=20, type="SAMAdapterbimodalMixModNewInTwinConvNEWwithcp"),
}


def make_input(cfg, batch=None, seed=None):
    """Synthetic RGB+LiDAR tensor: RGB ~ N(0,1); aux = sparse 5% U(0,1)."""
    b = batch or cfg["batch"]
    s = cfg["kwargs"]["img_size"]
    g = torch.Generator().manual_seed(cfg["in_seed"] if seed is None else seed)
    x = torch.randn(b, 6, s, s, generator=g)
    m = torch.rand(b, 3, s, s, generator=g) < 0.05
    x[:, 3:] = m.float() * torch.rand(b, 3, s, s, generator=g)
    return x


def weights_checksum(sd):
    return float(sum(v.double().abs().sum().item() for k, v in sorted(sd.items())))


def probe_index(numel, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (n,), generator=g)


# decode head: CFG-L decode_head dict (configs/DELIVER/..._RGBLIDAR_hard.py:57-66); `size` = side of f1 (img/4)
_HEAD_KW = dict(in_index=[0, 1, 2, 3], dropout_ratio=0.1, norm_cfg=dict(type="SyncBN", requires_grad=True),
                align_corners=False, loss_decode=dict(type="OhemCrossEntropy"))
HEAD_CONFIGS = {
    "head_tiny": dict(kwargs=dict(_HEAD_KW, in_channels=[64, 64, 64, 64], channels=64, num_classes=7), size=56, batch=2,
                      seed=11, in_seed=12),
    "head_odd": dict(kwargs=dict(_HEAD_KW, in_channels=[40, 40, 40, 40], channels=48, num_classes=25), size=40, batch=1,
                     seed=13, in_seed=14),
    "head_vitl": dict(kwargs=dict(_HEAD_KW, in_channels=[1024, 1024, 1024, 1024], channels=512, num_classes=25), size=256,
                      batch=1, seed=15, in_seed=16),
}


def make_head_inputs(cfg, batch=None, seed=None):
    """Seeded stand-ins for the backbone maps f1..f4: [B, C_i, size/2^i, size/2^i] ~ N(0,1)."""
    b = batch or cfg["batch"]
    g = torch.Generator().manual_seed(cfg["in_seed"] if seed is None else seed)
    return [torch.randn(b, c, cfg["size"] >> i, cfg["size"] >> i, generator=g) for i, c in enumerate(cfg["kwargs"]["in_channels"])]


def toy_encode_decode(num_classes, seed):
    """Fixed seeded stand-in for encode_decode used to pin slide_inference (reference method vs oracle): a 1x1 conv on the
    4x-downsampled crop, resized back to the crop size (the shape flow of encode_decode, ED:88-94)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(num_classes, 6, 1, 1, generator=g)

    def fn(crop):
        low = F.avg_pool2d(crop, 4)
        return F.interpolate(F.conv2d(low, w), size=crop.shape[2:], mode="bilinear", align_corners=False)
    return fn


def fake_convnext_checkpoint(twin_keys_shapes, seed=41):
    """A single-stream ConvNeXt state dict (mmpretrain key names: downsample_layers.*, stages.*, norm{i}.*) with seeded values,
    derived from the x-stream keys/shapes of a TwinConvNeXt: what TwinConvNeXt.init_weights (TC:403-443) duplicates into both streams."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in sorted(twin_keys_shapes):
        first, rest = k.split(".", 1)
        if first.endswith("_x"):
            base = first[:-2]
        elif first.startswith("norm_x"):
            base = "norm" + first[len("norm_x"):]
        else:
            continue
        sd[base + "." + rest] = torch.randn(shp, generator=g) * 0.1
    return sd


def fake_sam_checkpoint(vit_keys_shapes, seed=51, drop=True):
    """A seeded SAM image-encoder state dict (pos_embed, patch_embed.proj.*, blocks.N.*) for the given key/shape list.  With `drop`:
    block 2 is missing entirely, blocks.1.attn.rel_pos_h has a wrong length (a 63-row table) and an unexpected 'neck.0.weight' is
    present -- the three irregularities a non-strict load has to survive (checkpoint.py:343-360, load_state_dict :44-113)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in vit_keys_shapes:
        if drop and k.startswith("blocks.2."):
            continue
        if drop and k == "blocks.1.attn.rel_pos_h":
            shp = (63, shp[1])
        sd[k] = torch.randn(shp, generator=g) * 0.1
    if drop:
        sd["neck.0.weight"] = torch.ones(4, 4)
    return sd
