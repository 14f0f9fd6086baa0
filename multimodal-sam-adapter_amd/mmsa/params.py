"""Parameter tree of the backbone: the reference's state_dict keys, shapes and registration order
(IE:235-276, BK:54-99, AM:297-363,861-907, TC:296-380, ops/modules/ms_deform_attn.py:55-58),
so that checkpoints saved from the reference load with `load_state_dict` / mmcv `load_checkpoint` unchanged.
Only parameter *holders* live here -- the arithmetic is in the HIP library."""
import math

import torch
import torch.nn as nn

CONVNEXT_ARCH = {  # TC:185-226
    "atto": dict(depths=[2, 2, 6, 2], channels=[40, 80, 160, 320]),
    "femto": dict(depths=[2, 2, 6, 2], channels=[48, 96, 192, 384]),
    "pico": dict(depths=[2, 2, 6, 2], channels=[64, 128, 256, 512]),
    "nano": dict(depths=[2, 2, 8, 2], channels=[80, 160, 320, 640]),
    "tiny": dict(depths=[3, 3, 9, 3], channels=[96, 192, 384, 768]),
    "small": dict(depths=[3, 3, 27, 3], channels=[96, 192, 384, 768]),
    "base": dict(depths=[3, 3, 27, 3], channels=[128, 256, 512, 1024]),
    "large": dict(depths=[3, 3, 27, 3], channels=[192, 384, 768, 1536]),
    "xlarge": dict(depths=[3, 3, 27, 3], channels=[256, 512, 1024, 2048]),
    "huge": dict(depths=[3, 3, 27, 3], channels=[352, 704, 1408, 2816]),
}


class Node(nn.Module):
    """Plain container; children/parameters are attached by dotted name."""


def _attach(root, name, tensor, buffer=False):
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


def param_spec(cfg):
    """Ordered list of (name, shape, kind) with kind in {'param','buffer','counter'}."""
    D, depth, heads = cfg["embed_dim"], cfg["depth"], cfg["num_heads"]
    hd = D // heads
    patch = cfg["patch_size"]
    grid = cfg["pretrained_size"] // patch
    ws = cfg["window_size"]
    hidden = int(D * cfg["mlp_ratio"])
    spec = []

    def P(n, *s):
        spec.append((n, tuple(s), "param"))

    P("pos_embed", 1, grid, grid, D)
    P("level_embed", 3, D)
    P("patch_embed.proj.weight", D, 3, patch, patch)
    P("patch_embed.proj.bias", D)
    for i in range(depth):
        L = 2 * (grid if i in cfg["global_attn_indexes"] else ws) - 1
        b = f"blocks.{i}."
        P(b + "norm1.weight", D); P(b + "norm1.bias", D)
        if cfg.get("use_rel_pos", True):
            P(b + "attn.rel_pos_h", L, hd); P(b + "attn.rel_pos_w", L, hd)
        P(b + "attn.qkv.weight", 3 * D, D)
        if cfg.get("qkv_bias", True):      # IE:317
            P(b + "attn.qkv.bias", 3 * D)
        P(b + "attn.proj.weight", D, D); P(b + "attn.proj.bias", D)
        P(b + "norm2.weight", D); P(b + "norm2.bias", D)
        P(b + "mlp.lin1.weight", hidden, D); P(b + "mlp.lin1.bias", hidden)
        P(b + "mlp.lin2.weight", D, hidden); P(b + "mlp.lin2.bias", D)
    arch = cfg["arch"]
    arch = CONVNEXT_ARCH[arch] if isinstance(arch, str) else arch
    depths, chans = list(arch["depths"]), list(arch["channels"])
    t = "spm.twin_conv."
    for s in ("x", "y"):
        d = t + f"downsample_layers_{s}."
        P(d + "0.0.weight", chans[0], 3, 4, 4); P(d + "0.0.bias", chans[0])
        P(d + "0.1.weight", chans[0]); P(d + "0.1.bias", chans[0])
        for i in range(1, 4):
            P(d + f"{i}.0.weight", chans[i - 1]); P(d + f"{i}.0.bias", chans[i - 1])
            P(d + f"{i}.1.weight", chans[i], chans[i - 1], 2, 2); P(d + f"{i}.1.bias", chans[i])
    for s in ("x", "y"):
        for i in range(4):
            c = chans[i]
            for j in range(depths[i]):
                b = t + f"stages_{s}.{i}.{j}."
                P(b + "gamma", c)
                P(b + "depthwise_conv.weight", c, 1, 7, 7); P(b + "depthwise_conv.bias", c)
                P(b + "norm.weight", c); P(b + "norm.bias", c)
                P(b + "pointwise_conv1.weight", 4 * c, c); P(b + "pointwise_conv1.bias", 4 * c)
                P(b + "pointwise_conv2.weight", c, 4 * c); P(b + "pointwise_conv2.bias", c)
    for i in range(4):
        for s in ("x", "y"):
            P(t + f"norm_{s}{i}.weight", chans[i]); P(t + f"norm_{s}{i}.bias", chans[i])
    inpl = cfg["conv_inplane"]
    nch = [4 * inpl, 8 * inpl, 16 * inpl, 32 * inpl]
    for i in range(4):
        P(f"spm.fc{i+1}.weight", D, nch[i], 1, 1); P(f"spm.fc{i+1}.bias", D)
    f = "spm.smart_fusion."
    for i in range(4):
        P(f + f"enhance_blocks.{i}.conv_atten.conv.weight", nch[i], nch[i], 1, 1)
        P(f + f"enhance_blocks.{i}.conv_atten.gn.weight", nch[i]); P(f + f"enhance_blocks.{i}.conv_atten.gn.bias", nch[i])
    for m in ("rgb", "sne"):
        for i in range(4):
            c = nch[i] // 2
            b = f + f"global_feature_encoder_{m}.{i}."
            P(b + "norm1.body.weight", c); P(b + "norm1.body.bias", c)
            P(b + "attn.scale", 8, 1, 1); P(b + "attn.scale2")
            P(b + "attn.qkv1.weight", 3 * c, c // 32, 1, 1)
            P(b + "attn.qkv2.weight", 3 * c, 3 * c // 32, 3, 3)
            P(b + "attn.proj.weight", c, c, 1, 1)
    for m in ("rgb", "sne"):
        for i in range(4):
            c = nch[i] // 2
            b = f + f"local_feature_encoder_{m}.{i}."
            P(b + "scale")
            P(b + "bottleneckBlock.0.weight", 2 * c, c, 1, 1)
            P(b + "bottleneckBlock.2.weight", 2 * c, 1, 3, 3)
            P(b + "bottleneckBlock.4.weight", c, 2 * c, 1, 1)
    for i in range(4):
        c = nch[i]
        mip = max(8, c // 32)
        b = f + f"ca_blocks.{i}.coord_atten."
        P(b + "conv1.weight", mip, c, 1, 1); P(b + "conv1.bias", mip)
        P(b + "bn1.weight", mip); P(b + "bn1.bias", mip)
        spec.append((b + "bn1.running_mean", (mip,), "buffer"))
        spec.append((b + "bn1.running_var", (mip,), "buffer"))
        spec.append((b + "bn1.num_batches_tracked", (), "counter"))
        P(b + "conv_h.weight", c, mip, 1, 1); P(b + "conv_h.bias", c)
        P(b + "conv_w.weight", c, mip, 1, 1); P(b + "conv_w.bias", c)
    img = cfg["img_size"]
    for i in range(4):
        s = img // 2 ** (i + 2)
        b = f + f"fuse_blocks.{i}."
        P(b + "gammax.scale"); P(b + "gammay.scale")
        P(b + "norm.weight", s * s); P(b + "norm.bias", s * s)
    for i in range(4):
        P(f + f"scale_layers.{i}.scale1"); P(f + f"scale_layers.{i}.scale2")
    for i in range(4):
        c = nch[i]
        b = f + f"detail_feature_extractions.{i}."
        P(b + "project_in.weight", 2 * c, c, 1, 1)
        P(b + "dwconv.weight", 2 * c, 2, 3, 3)
        P(b + "project_out.weight", c, c, 1, 1)
    P("up.weight", D, D, 2, 2); P("up.bias", D)
    M, Pn = cfg["deform_num_heads"], cfg["n_points"]
    dv = int(D * cfg["deform_ratio"])
    hid = int(D * cfg["cffn_ratio"])

    def msda(b, L):
        P(b + "sampling_offsets.weight", M * L * Pn * 2, D); P(b + "sampling_offsets.bias", M * L * Pn * 2)
        P(b + "attention_weights.weight", M * L * Pn, D); P(b + "attention_weights.bias", M * L * Pn)
        P(b + "value_proj.weight", dv, D); P(b + "value_proj.bias", dv)
        P(b + "output_proj.weight", D, dv); P(b + "output_proj.bias", D)

    def extractor(b):
        P(b + "query_norm.weight", D); P(b + "query_norm.bias", D)
        P(b + "feat_norm.weight", D); P(b + "feat_norm.bias", D)
        msda(b + "attn.", 1)
        if cfg.get("with_cffn", True):   # AM:485-488: registered only then
            P(b + "ffn.fc1.weight", hid, D); P(b + "ffn.fc1.bias", hid)
            P(b + "ffn.dwconv.dwconv.weight", hid, 1, 3, 3); P(b + "ffn.dwconv.dwconv.bias", hid)
            P(b + "ffn.fc2.weight", D, hid); P(b + "ffn.fc2.bias", D)
            P(b + "ffn_norm.weight", D); P(b + "ffn_norm.bias", D)

    n_int = len(cfg["interaction_indexes"])
    for i in range(n_int):
        b = f"interactions.{i}."
        P(b + "injector.gamma", D)
        P(b + "injector.query_norm.weight", D); P(b + "injector.query_norm.bias", D)
        P(b + "injector.feat_norm.weight", D); P(b + "injector.feat_norm.bias", D)
        msda(b + "injector.attn.", 3)
        extractor(b + "extractor.")
        if i == n_int - 1 and cfg["use_extra_extractor"]:
            extractor(b + "extra_extractors.0.")
            extractor(b + "extra_extractors.1.")
    for i in range(1, 5):
        P(f"norm{i}.weight", D); P(f"norm{i}.bias", D)
        spec.append((f"norm{i}.running_mean", (D,), "buffer"))
        spec.append((f"norm{i}.running_var", (D,), "buffer"))
        spec.append((f"norm{i}.num_batches_tracked", (), "counter"))
    return spec


def build_tree(root, cfg):
    """Attach zero-initialised parameters/buffers following param_spec, then apply the default init."""
    for name, shape, kind in param_spec(cfg):
        if kind == "param":
            _attach(root, name, torch.zeros(shape))
        elif kind == "buffer":
            _attach(root, name, torch.zeros(shape), buffer=True)
        else:
            _attach(root, name, torch.zeros((), dtype=torch.long), buffer=True)
    default_init(root, cfg)


@torch.no_grad()
def default_init(root, cfg):
    """Default initialisation with the reference's distributions (BK:119-134,145-147; IE:296;
    ops/modules/ms_deform_attn.py:64-81; AM:238-239,292,523; TC:390-397).  Exact RNG streams are not part of the
    contract (checkpoints are loaded over it)."""
    sd = dict(root.named_parameters())
    sd.update(dict(root.named_buffers()))
    M, Pn = cfg["deform_num_heads"], cfg["n_points"]
    for k, v in sd.items():
        leaf = k.split(".")[-1]
        if leaf == "num_batches_tracked":
            continue
        if leaf == "running_var":
            v.fill_(1.0)
        elif leaf == "running_mean":
            v.zero_()
        elif "sampling_offsets" in k:
            if leaf == "weight":
                v.zero_()
            else:
                L = v.numel() // (M * Pn * 2)
                th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
                g = torch.stack([th.cos(), th.sin()], -1)
                g = (g / g.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, Pn, 1)
                for i in range(Pn):
                    g[:, :, i, :] *= i + 1
                v.copy_(g.view(-1))
        elif "attention_weights" in k:
            v.zero_()
        elif ("value_proj" in k or "output_proj" in k) and leaf == "weight":
            nn.init.xavier_uniform_(v)
        elif leaf in ("rel_pos_h", "rel_pos_w") or k == "pos_embed":
            v.zero_()
        elif k == "level_embed":
            v.normal_()
        elif k.endswith("injector.gamma"):
            v.fill_(cfg["init_values"])
        elif leaf == "gamma":
            v.fill_(1.0)  # ConvNeXt layer_scale_init_value=1.0 (AM:881)
        elif leaf in ("scale1", "scale2"):
            v.fill_(1.0)
        elif leaf == "scale" and v.ndim == 0:
            v.zero_()  # GFFM gammax/gammay, MobileNetV2.scale
        elif leaf == "scale":
            v.fill_(1.0)  # AttentionBase temperature
        elif leaf == "bias":
            v.zero_()
        elif leaf == "weight" and v.ndim == 1:
            v.fill_(1.0)
        elif leaf == "weight" and v.ndim == 2:
            nn.init.trunc_normal_(v, std=0.02)
        elif leaf == "weight" and v.ndim == 4:
            if k == "up.weight":
                fan_out = v.shape[2] * v.shape[3] * v.shape[1]
            else:
                fan_out = v.shape[2] * v.shape[3] * v.shape[0]
            v.normal_(0, math.sqrt(2.0 / max(fan_out, 1)))
        else:
            v.zero_()
