"""The build's deterministic "live weights" generator (SURVEY 8c fixture hygiene).  Test infrastructure shared by the golden
generator (tools/oracle/make_golden.py, which applies it to the imported reference), the tests and smoke(): data only, no
reference code."""
import math

import torch


def seeded_state_dict(model, seed=0):
    """The build's deterministic 'live weights' generator (SURVEY 8c fixture hygiene): every
    parameter and buffer is overwritten with seeded non-trivial values so no branch is dead
    (injector gamma, GFFM gammas, MobileNetV2 scale, rel_pos, BN running stats, MSDA
    offset/attention weights).  Keys are visited in sorted order so the same recipe can be
    applied to the reference model via load_state_dict."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    out = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        shp = tuple(v.shape)

        def rn(std=1.0):
            return torch.randn(shp, generator=g) * std
        leaf = k.split(".")[-1]
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros((), dtype=torch.long)
        elif leaf == "running_var":
            out[k] = 0.5 + torch.rand(shp, generator=g)
        elif leaf == "running_mean":
            out[k] = rn(0.1)
        elif "sampling_offsets" in k and leaf == "bias":
            out[k] = v.clone()  # keep ring init (+-1..+-4 px)
        elif "sampling_offsets" in k and leaf == "weight":
            out[k] = rn(0.05)
        elif "attention_weights" in k:
            out[k] = rn(0.05) if leaf == "weight" else rn(0.5)
        elif leaf in ("rel_pos_h", "rel_pos_w"):
            out[k] = rn(0.05)
        elif k in ("pos_embed", "level_embed"):
            out[k] = rn(0.3)
        elif leaf == "gamma" and "injector" in k:
            out[k] = 0.5 + 0.2 * torch.rand(shp, generator=g)
        elif leaf == "gamma":  # ConvNeXt layer scale
            out[k] = 0.2 + 0.2 * torch.rand(shp, generator=g)
        elif leaf in ("scale", "scale1", "scale2") and v.ndim == 0:
            out[k] = torch.tensor(0.6) + 0.3 * torch.rand((), generator=g)
        elif leaf == "scale":  # AttentionBase per-head temperature [8,1,1]
            out[k] = 0.8 + 0.4 * torch.rand(shp, generator=g)
        elif leaf == "weight" and v.ndim == 1:  # norm weights
            out[k] = 1.0 + rn(0.1)
        elif leaf == "bias":
            out[k] = rn(0.05)
        elif leaf == "weight":
            fan_in = v[0].numel()
            out[k] = rn(0.7 / math.sqrt(max(fan_in, 1)))
        else:
            out[k] = rn(0.1)
        out[k] = out[k].to(v.dtype)
    return out


def peaky_attention(sd, embed_dim, factor, blocks=None):
    """`sd` with the q and k rows of every attn.qkv projection (weight and bias, the first 2 * embed_dim rows: IE:488 memory order) scaled
    by `factor`: attention logits factor^2 times larger, everything else unchanged -- the fixtures with peaky attention (released SAM
    checkpoints are peakier than the seeded generator above) are captured from the reference with these weights.  `blocks`: only these
    block indices (a MIXED state: some blocks on single fp16 operands, the others on pairs), None = all."""
    out = dict(sd)
    for k in sd:
        if blocks is not None and not any(k.startswith(f"blocks.{i}.") for i in blocks):
            continue
        if k.endswith("attn.qkv.weight") or k.endswith("attn.qkv.bias"):
            v = sd[k].clone()
            v[:2 * embed_dim] *= factor
            out[k] = v
    return out


def large_magnitude(sd, spec):
    """`sd` re-parametrised so that intermediate tensors of the ViT blocks leave the range of the fp16-based operand formats (+-57344 / +-65504) while the
    function the model computes stays an ordinary one -- the reference computes in fp32 and is indifferent.  `spec` = dict with
      ln2: (block, channel)  norm2.weight[channel] x 3e4 and column `channel` of mlp.lin1.weight / 3e4: the post-LayerNorm plane that feeds lin1 holds ~1e5 there;
      ln1: (block, channel)  the same for norm1 and attn.qkv;
      gelu: (block, unit)    mlp.lin1.bias[unit] = 6e4 and column `unit` of mlp.lin2.weight x 1e-4: the GELU hidden activation of that unit is ~6e4 in every token.
    The fixtures `*_wide` are captured from the imported reference with these weights (tools/oracle/make_golden.py)."""
    out = dict(sd)

    def mod(key):
        out[key] = out[key].clone()
        return out[key]
    if "ln2" in spec:
        b, c = spec["ln2"]
        mod(f"blocks.{b}.norm2.weight")[c] *= 3.0e4
        mod(f"blocks.{b}.mlp.lin1.weight")[:, c] /= 3.0e4
    if "ln1" in spec:
        b, c = spec["ln1"]
        mod(f"blocks.{b}.norm1.weight")[c] *= 3.0e4
        mod(f"blocks.{b}.attn.qkv.weight")[:, c] /= 3.0e4
    if "gelu" in spec:
        b, j = spec["gelu"]
        mod(f"blocks.{b}.mlp.lin1.bias")[j] = 6.0e4
        mod(f"blocks.{b}.mlp.lin2.weight")[:, j] *= 1.0e-4
    return out
