"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the MM-SAM-Adapter
image-encoder forward in plain PyTorch fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (multimodal-sam-adapter_amd/mmsa) never does; it fails loudly without the
HIP library.

Parity pinning: this restatement is checked in tests/test_oracle_golden.py against golden
vectors produced by importing the UNMODIFIED reference in the build container
(tools/oracle/make_golden.py -> tests/golden/*.npz).  All citations are relative to
/root/reference/segmentation/:
  BK = mmseg_custom/models/backbones/image_encoder_adapter_bimodal_mix_mod_new_in_twin_convnext_new.py
  IE = mmseg_custom/models/backbones/base/image_encoder.py
  AM = mmseg_custom/models/backbones/adapter_modules_multimodal_mix_mod_new_in_twin_convnext_new.py
  TC = mmseg_custom/models/backbones/base/twin_convnext.py
  OPS = ops/
Module/parameter names are the reference's so that state_dict keys are identical.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

CONVNEXT_ARCH = {  # TC:185-226
    "tiny": dict(depths=[3, 3, 9, 3], channels=[96, 192, 384, 768]),
    "small": dict(depths=[3, 3, 27, 3], channels=[96, 192, 384, 768]),
    "base": dict(depths=[3, 3, 27, 3], channels=[128, 256, 512, 1024]),
    "large": dict(depths=[3, 3, 27, 3], channels=[192, 384, 768, 1536]),
}


# ----------------------------------------------------------------------------- MSDA core
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """Restates OPS/functions/ms_deform_attn_func.py:53-75 (grid_sample form), which
    OPS/test.py:26-75 pins as equivalent to the CUDA kernel OPS/src/cuda/ms_deform_im2col_cuda.cuh:237-299."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    vals = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for l, (h, w) in enumerate(spatial_shapes):
        h, w = int(h), int(w)
        v = vals[l].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def msda_direct(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """Second, independent restatement that follows the CUDA kernel literally
    (OPS/src/cuda/ms_deform_im2col_cuda.cuh:33-84 bilinear, :237-299 loop), vectorised in torch.
    Used to cross-check msda_core and as the model for oracle/msda_ref.c."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = value.new_zeros(N, Lq, M, D)
    bidx = torch.arange(N).view(N, 1, 1, 1).expand(N, Lq, M, P)
    midx = torch.arange(M).view(1, 1, M, 1).expand(N, Lq, M, P)
    for l in range(L):
        H, W = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
        start = int(level_start_index[l])
        loc = sampling_locations[:, :, :, l]  # N,Lq,M,P,2
        w_im = loc[..., 0] * W - 0.5
        h_im = loc[..., 1] * H - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low = torch.floor(h_im)
        w_low = torch.floor(w_im)
        lh = h_im - h_low
        lw = w_im - w_low
        hh, hw = 1 - lh, 1 - lw
        h_low = h_low.long()
        w_low = w_low.long()
        acc = value.new_zeros(N, Lq, M, P, D)
        for dh, dw, wt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            hi = h_low + dh
            wi = w_low + dw
            ok = inside & (hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1)
            idx = start + hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)
            v = value[bidx, idx, midx]  # N,Lq,M,P,D
            acc = acc + v * (wt * ok.to(value.dtype)).unsqueeze(-1)
        out = out + (acc * attention_weights[:, :, :, l].unsqueeze(-1)).sum(3)
    return out.view(N, Lq, M * D)


class MSDeformAttn(nn.Module):
    """OPS/modules/ms_deform_attn.py:28-130."""

    def __init__(self, d_model, n_levels, n_heads, n_points, ratio):
        super().__init__()
        self.d_model, self.n_levels, self.n_heads, self.n_points, self.ratio = d_model, n_levels, n_heads, n_points, ratio
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, int(d_model * ratio))
        self.output_proj = nn.Linear(int(d_model * ratio), d_model)
        self._reset_parameters()

    def _reset_parameters(self):  # OPS/modules/ms_deform_attn.py:64-81
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, spatial_shapes, level_start_index):
        N, Lq, _ = query.shape
        _, Lin, _ = input_flatten.shape
        value = self.value_proj(input_flatten).view(N, Lin, self.n_heads, int(self.ratio * self.d_model) // self.n_heads)
        off = self.sampling_offsets(query).view(N, Lq, self.n_heads, self.n_levels, self.n_points, 2)
        aw = self.attention_weights(query).view(N, Lq, self.n_heads, self.n_levels * self.n_points)
        aw = F.softmax(aw, -1).view(N, Lq, self.n_heads, self.n_levels, self.n_points)
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        out = msda_core(value, spatial_shapes, loc, aw)
        return self.output_proj(out)


# ----------------------------------------------------------------------------- SAM ViT (IE)
def window_partition(x, ws):  # IE:504-526
    B, H, W, C = x.shape
    ph = (ws - H % ws) % ws
    pw = (ws - W % ws) % ws
    if ph > 0 or pw > 0:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):  # IE:529-551
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    if Hp > H or Wp > W:
        x = x[:, :H, :W, :].contiguous()
    return x


def rel_pos_index(q_size, k_size):
    """Integer index table of IE:579-584 (float coords -> .long())."""
    q = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    return ((q - k) + (k_size - 1) * max(q_size / k_size, 1.0)).long()


def get_rel_pos(q_size, k_size, rel_pos):  # IE:554-584
    L = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != L:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=L, mode="linear")
        r = r.reshape(-1, L).permute(1, 0)
    else:
        r = rel_pos
    return r[rel_pos_index(q_size, k_size)]


def add_decomposed_rel_pos(attn, q, rph, rpw, q_size, k_size):  # IE:587-623
    qh, qw = q_size
    kh, kw = k_size
    Rh = get_rel_pos(qh, kh, rph)
    Rw = get_rel_pos(qw, kw, rpw)
    B, _, dim = q.shape
    rq = q.reshape(B, qh, qw, dim)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    return (attn.view(B, qh, qw, kh, kw) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(B, qh * qw, kh * kw)


class Attention(nn.Module):  # IE:426-501
    def __init__(self, dim, num_heads, input_size, qkv_bias=True, use_rel_pos=True):
        super().__init__()
        self.num_heads = num_heads
        hd = dim // num_heads
        self.scale = hd ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)   # IE:317
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if use_rel_pos:   # IE:320-327: the tables exist only then
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, hd))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, hd))

    def forward(self, x):
        B, H, W, _ = x.shape
        qkv = self.qkv(x).reshape(B, H * W, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.reshape(3, B * self.num_heads, H * W, -1).unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        if self.use_rel_pos:
            attn = add_decomposed_rel_pos(attn, q, self.rel_pos_h, self.rel_pos_w, (H, W), (H, W))
        attn = attn.softmax(dim=-1)
        x = (attn @ v).view(B, self.num_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
        return self.proj(x)


class MLPBlock(nn.Module):  # IE:154-167
    def __init__(self, dim, hidden):
        super().__init__()
        self.lin1 = nn.Linear(dim, hidden)
        self.lin2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.lin2(F.gelu(self.lin1(x)))


class Block(nn.Module):  # IE:331-423
    def __init__(self, dim, num_heads, mlp_ratio, window_size, input_size, qkv_bias=True, use_rel_pos=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, input_size if window_size == 0 else (window_size, window_size), qkv_bias, use_rel_pos)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = MLPBlock(dim, int(dim * mlp_ratio))
        self.window_size = window_size

    def forward(self, x, H, W):
        x = x.unflatten(1, (H, W))
        shortcut = x
        x = self.norm1(x)
        if self.window_size > 0:
            x, pad_hw = window_partition(x, self.window_size)
        x = self.attn(x)
        if self.window_size > 0:
            x = window_unpartition(x, self.window_size, pad_hw, (H, W))
        x = shortcut + x
        x = x + self.mlp(self.norm2(x))
        return x.flatten(1, 2)


class PatchEmbed(nn.Module):  # IE:626-671
    def __init__(self, patch, in_chans, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        x = self.proj(x)
        Hp, Wp = x.shape[2], x.shape[3]
        return x.permute(0, 2, 3, 1).flatten(1, 2), Hp, Wp


# ----------------------------------------------------------------------------- TwinConvNeXt (TC)
class LN2d(nn.LayerNorm):  # mmpretrain_custom/models/utils/norm.py:51-90
    def forward(self, x, channel_last=False):
        if channel_last:
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        x = F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2).contiguous()


class ConvNeXtBlock(nn.Module):  # TC:23-132
    def __init__(self, c, layer_scale):
        super().__init__()
        self.gamma = nn.Parameter(layer_scale * torch.ones(c))
        self.depthwise_conv = nn.Conv2d(c, c, kernel_size=7, padding=3, groups=c)
        self.norm = LN2d(c, eps=1e-6)
        self.pointwise_conv1 = nn.Linear(c, 4 * c)
        self.pointwise_conv2 = nn.Linear(4 * c, c)

    def forward(self, x):
        sc = x
        x = self.depthwise_conv(x).permute(0, 2, 3, 1)
        x = self.norm(x, channel_last=True)
        x = self.pointwise_conv2(F.gelu(self.pointwise_conv1(x))).permute(0, 3, 1, 2)
        x = x.mul(self.gamma.view(1, -1, 1, 1))
        return sc + x


class TwinConvNeXt(nn.Module):  # TC:136-476
    def __init__(self, arch):
        super().__init__()
        a = CONVNEXT_ARCH[arch] if isinstance(arch, str) else arch
        self.depths, self.channels = list(a["depths"]), list(a["channels"])
        # registration order TC:296-374: downsample_layers_x, downsample_layers_y, stages_x, stages_y
        for s in ("x", "y"):
            ds = nn.ModuleList()
            ds.append(nn.Sequential(nn.Conv2d(3, self.channels[0], kernel_size=4, stride=4), LN2d(self.channels[0], eps=1e-6)))
            for i in range(1, 4):
                ds.append(nn.Sequential(LN2d(self.channels[i - 1], eps=1e-6),
                                        nn.Conv2d(self.channels[i - 1], self.channels[i], kernel_size=2, stride=2)))
            setattr(self, f"downsample_layers_{s}", ds)
        for s in ("x", "y"):
            stages = nn.ModuleList()
            for i in range(4):
                stages.append(nn.Sequential(*[ConvNeXtBlock(self.channels[i], 1.0) for _ in range(self.depths[i])]))
            setattr(self, f"stages_{s}", stages)
        # TC:376-380 registers norm_x{i}, norm_y{i} interleaved per stage
        for i in range(4):
            self.add_module(f"norm_x{i}", LN2d(self.channels[i], eps=1e-6))
            self.add_module(f"norm_y{i}", LN2d(self.channels[i], eps=1e-6))

    def _stream(self, x, s):
        outs = []
        for i in range(4):
            x = getattr(self, f"downsample_layers_{s}")[i](x)
            x = getattr(self, f"stages_{s}")[i](x)
            outs.append(getattr(self, f"norm_{s}{i}")(x))
        return outs

    def forward(self, x, y):  # TC:445-476
        return [torch.cat((a, b), dim=1) for a, b in zip(self._stream(x, "x"), self._stream(y, "y"))]


# ----------------------------------------------------------------------------- RoadFormer2Neck (AM)
class _Body(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class WithBiasLN(nn.Module):  # AM:51-74 (per-pixel LN over channels, biased var, eps 1e-5)
    def __init__(self, c):
        super().__init__()
        self.body = _Body(c)

    def forward(self, x):
        b, c, h, w = x.shape
        t = x.flatten(2).transpose(1, 2)
        mu = t.mean(-1, keepdim=True)
        var = t.var(-1, keepdim=True, unbiased=False)
        t = (t - mu) / torch.sqrt(var + 1e-5) * self.body.weight + self.body.bias
        return t.transpose(1, 2).reshape(b, c, h, w)


class AttentionBase(nn.Module):  # AM:75-109
    def __init__(self, dim, num_heads=8, groups=32):
        super().__init__()
        self.num_heads = num_heads
        self.scale = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.scale2 = nn.Parameter(torch.tensor(1.0))
        self.qkv1 = nn.Conv2d(dim, dim * 3, kernel_size=1, groups=groups, bias=False)
        self.qkv2 = nn.Conv2d(dim * 3, dim * 3, kernel_size=3, padding=1, groups=groups, bias=False)
        self.proj = nn.Conv2d(dim, dim, kernel_size=1, bias=False)

    def forward(self, x):
        b, c, h, w = x.shape
        q, k, v = self.qkv2(self.qkv1(x)).chunk(3, dim=1)
        nh = self.num_heads
        q = F.normalize(q.reshape(b, nh, c // nh, h * w), dim=-1)
        k = F.normalize(k.reshape(b, nh, c // nh, h * w), dim=-1)
        v = v.reshape(b, nh, c // nh, h * w)
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        out = (attn @ v).reshape(b, c, h, w)
        return x + self.proj(out) * self.scale2


class GFE(nn.Module):  # AM:133-145
    def __init__(self, dim):
        super().__init__()
        self.norm1 = WithBiasLN(dim)
        self.attn = AttentionBase(dim)

    def forward(self, x):
        return x + self.attn(self.norm1(x))


class MobileNetV2(nn.Module):  # AM:281-295
    def __init__(self, c):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(0.0))
        self.bottleneckBlock = nn.Sequential(
            nn.Conv2d(c, 2 * c, 1, bias=False), nn.ReLU6(),
            nn.Conv2d(2 * c, 2 * c, 3, padding=1, groups=2 * c, bias=False), nn.ReLU6(),
            nn.Conv2d(2 * c, c, 1, bias=False))

    def forward(self, x):
        return self.bottleneckBlock(x) * self.scale + x


class Scale(nn.Module):
    def __init__(self, v):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(float(v)))


class GFFM(nn.Module):  # AM:234-267
    def __init__(self, hw):
        super().__init__()
        self.gammax = Scale(0)
        self.gammay = Scale(0)
        self.norm = nn.LayerNorm(hw)

    def forward(self, g):
        c = g.size(1) // 2
        x, y = torch.split(g, (c, c), dim=1)
        b, _, h, w = x.shape
        xf, yf = x.reshape(b, c, -1), y.reshape(b, c, -1)
        ax = F.softmax(torch.bmm(xf, yf.permute(0, 2, 1)), dim=-1)
        ay = F.softmax(torch.bmm(yf, xf.permute(0, 2, 1)), dim=-1)
        ox = self.gammax.scale * torch.bmm(ax, yf) + xf
        oy = self.gammay.scale * torch.bmm(ay, xf) + yf
        out = self.norm(torch.cat((ox, oy), dim=1))
        return out.view(b, 2 * c, h, w)


class GatedMlp(nn.Module):  # AM:110-132 (ffn_expansion_factor=1, bias=False)
    def __init__(self, c):
        super().__init__()
        self.project_in = nn.Conv2d(c, 2 * c, 1, bias=False)
        self.dwconv = nn.Conv2d(2 * c, 2 * c, 3, padding=1, groups=c, bias=False)
        self.project_out = nn.Conv2d(c, c, 1, bias=False)

    def forward(self, x):
        x1, x2 = self.dwconv(self.project_in(x)).chunk(2, dim=1)
        return self.project_out(F.gelu(x1) * x2)


class _ConvGN(nn.Module):  # mmcv_custom/cnn/bricks/conv_module.py:71-212: conv -> GN(32) -> ReLU
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 1, bias=False)
        self.gn = nn.GroupNorm(32, c)

    def forward(self, x):
        return F.relu(self.gn(self.conv(x)))


class FFRM(nn.Module):  # AM:148-162
    def __init__(self, c):
        super().__init__()
        self.conv_atten = _ConvGN(c)

    def forward(self, x):
        a = torch.sigmoid(self.conv_atten(F.avg_pool2d(x, x.shape[2:])))
        return x + x * a


class Scale2(nn.Module):  # AM:268-280
    def __init__(self):
        super().__init__()
        self.scale1 = nn.Parameter(torch.tensor(1.0))
        self.scale2 = nn.Parameter(torch.tensor(1.0))

    def forward(self, x, y):
        return x * self.scale1 + y * self.scale2


class CoordinateAttention(nn.Module):  # AM:176-201
    def __init__(self, c):
        super().__init__()
        mip = max(8, c // 32)
        self.conv1 = nn.Conv2d(c, mip, 1)
        self.bn1 = nn.BatchNorm2d(mip)  # SyncBatchNorm in eval == BatchNorm2d in eval
        self.conv_h = nn.Conv2d(mip, c, 1)
        self.conv_w = nn.Conv2d(mip, c, 1)

    def forward(self, x):
        n, c, h, w = x.shape
        xh = x.mean(3, keepdim=True)
        xw = x.mean(2, keepdim=True).permute(0, 1, 3, 2)
        y = self.bn1(self.conv1(torch.cat([xh, xw], dim=2)))
        y = y * (F.relu6(y + 3) / 6)
        yh, yw = torch.split(y, [h, w], dim=2)
        ah = self.conv_h(yh).sigmoid()
        aw = self.conv_w(yw.permute(0, 1, 3, 2)).sigmoid()
        return x * aw * ah


class CA(nn.Module):  # AM:202-221
    def __init__(self, c):
        super().__init__()
        self.coord_atten = CoordinateAttention(c)

    def forward(self, x):
        return x + self.coord_atten(x)


class RoadFormer2Neck(nn.Module):  # AM:297-394
    def __init__(self, chans, img):
        super().__init__()
        self.chans = chans
        self.enhance_blocks = nn.ModuleList([FFRM(c) for c in chans])
        self.global_feature_encoder_rgb = nn.ModuleList([GFE(c // 2) for c in chans])
        self.global_feature_encoder_sne = nn.ModuleList([GFE(c // 2) for c in chans])
        self.local_feature_encoder_rgb = nn.ModuleList([MobileNetV2(c // 2) for c in chans])
        self.local_feature_encoder_sne = nn.ModuleList([MobileNetV2(c // 2) for c in chans])
        self.ca_blocks = nn.ModuleList([CA(c) for c in chans])
        fuse, scales = [], []
        for i in range(4):
            s = img // 2 ** (i + 2)
            fuse.append(GFFM(s * s))
            scales.append(Scale2())
        # registration order AM:358-363: fuse_blocks, scale_layers, detail_feature_extractions
        self.fuse_blocks = nn.ModuleList(fuse)
        self.scale_layers = nn.ModuleList(scales)
        self.detail_feature_extractions = nn.ModuleList([GatedMlp(c) for c in chans])

    def forward(self, feats):
        outs = []
        for i, f in enumerate(feats):
            c = self.chans[i] // 2
            fr, fs = torch.split(f, (c, c), dim=1)
            g = torch.cat((self.global_feature_encoder_rgb[i](fr), self.global_feature_encoder_sne[i](fs)), dim=1)
            l = torch.cat((self.local_feature_encoder_rgb[i](fr), self.local_feature_encoder_sne[i](fs)), dim=1)
            g = self.enhance_blocks[i](self.fuse_blocks[i](g))
            l = self.detail_feature_extractions[i](l)
            outs.append(self.ca_blocks[i](self.scale_layers[i](g, l)))
        return outs


class SpatialPriorModuleBimodal(nn.Module):  # AM:861-964
    def __init__(self, inplanes, embed_dim, img_size, arch):
        super().__init__()
        self.twin_conv = TwinConvNeXt(arch)
        chans = [4 * inplanes, 8 * inplanes, 16 * inplanes, 32 * inplanes]
        self.fc1 = nn.Conv2d(chans[0], embed_dim, 1)
        self.fc2 = nn.Conv2d(chans[1], embed_dim, 1)
        self.fc3 = nn.Conv2d(chans[2], embed_dim, 1)
        self.fc4 = nn.Conv2d(chans[3], embed_dim, 1)
        self.smart_fusion = RoadFormer2Neck(chans, img_size)

    def forward(self, x, y, taps=None):
        feats = self.twin_conv(x, y)
        if taps is not None:
            for i, f in enumerate(feats):
                taps[f"twin{i}"] = f
        feats = self.smart_fusion(feats)
        if taps is not None:
            for i, f in enumerate(feats):
                taps[f"fuse{i}"] = f
        cs = [fc(f) for fc, f in zip((self.fc1, self.fc2, self.fc3, self.fc4), feats)]
        return [c.flatten(2).transpose(1, 2) for c in cs]


# ----------------------------------------------------------------------------- injector / extractor (AM)
def get_reference_points(shapes, dtype):  # AM:397-409
    refs = []
    for H, W in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=dtype),
                                torch.linspace(0.5, W - 0.5, W, dtype=dtype), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1)[None] / W, ry.reshape(-1)[None] / H), -1))
    return torch.cat(refs, 1)[:, :, None]


def deform_inputs(h, w, dtype=torch.float32):  # AM:412-431
    ss1 = torch.as_tensor([(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)], dtype=torch.long)
    lsi1 = torch.cat((ss1.new_zeros((1,)), ss1.prod(1).cumsum(0)[:-1]))
    ref1 = get_reference_points([(h // 16, w // 16)], dtype)
    ss2 = torch.as_tensor([(h // 16, w // 16)], dtype=torch.long)
    lsi2 = torch.cat((ss2.new_zeros((1,)), ss2.prod(1).cumsum(0)[:-1]))
    ref2 = get_reference_points([(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)], dtype)
    return [ref1, ss1, lsi1], [ref2, ss2, lsi2]


class _DW(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dwconv = nn.Conv2d(c, c, 3, 1, 1, bias=True, groups=c)


class ConvFFN(nn.Module):  # AM:434-471
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = _DW(hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, H, W):
        x = self.fc1(x)
        B, N, C = x.shape
        n = N // 21
        conv = self.dwconv.dwconv
        x1 = conv(x[:, 0:16 * n].transpose(1, 2).reshape(B, C, H * 2, W * 2)).flatten(2).transpose(1, 2)
        x2 = conv(x[:, 16 * n:20 * n].transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
        x3 = conv(x[:, 20 * n:].transpose(1, 2).reshape(B, C, H // 2, W // 2)).flatten(2).transpose(1, 2)
        return self.fc2(F.gelu(torch.cat([x1, x2, x3], dim=1)))


class Extractor(nn.Module):  # AM:474-511
    def __init__(self, dim, heads, n_points, ratio, cffn_ratio, with_cffn=True):
        super().__init__()
        self.query_norm = nn.LayerNorm(dim, eps=1e-6)
        self.feat_norm = nn.LayerNorm(dim, eps=1e-6)
        self.attn = MSDeformAttn(dim, 1, heads, n_points, ratio)
        self.with_cffn = with_cffn
        if with_cffn:   # AM:485-488: the ConvFFN and its norm exist only then
            self.ffn = ConvFFN(dim, int(dim * cffn_ratio))
            self.ffn_norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, query, ref, feat, ss, lsi, H, W):
        query = query + self.attn(self.query_norm(query), ref, self.feat_norm(feat), ss, lsi)
        if not self.with_cffn:   # AM:499-500
            return query
        return query + self.ffn(self.ffn_norm(query), H, W)


class Injector(nn.Module):  # AM:514-542
    def __init__(self, dim, heads, n_points, ratio, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))
        self.query_norm = nn.LayerNorm(dim, eps=1e-6)
        self.feat_norm = nn.LayerNorm(dim, eps=1e-6)
        self.attn = MSDeformAttn(dim, 3, heads, n_points, ratio)

    def forward(self, query, ref, feat, ss, lsi):
        return query + self.gamma * self.attn(self.query_norm(query), ref, self.feat_norm(feat), ss, lsi)


class InteractionBlock(nn.Module):  # AM:545-581
    def __init__(self, dim, heads, n_points, ratio, cffn_ratio, init_values, extra, with_cffn=True):
        super().__init__()
        self.injector = Injector(dim, heads, n_points, ratio, init_values)
        self.extractor = Extractor(dim, heads, n_points, ratio, cffn_ratio, with_cffn)
        self.extra_extractors = nn.Sequential(*[Extractor(dim, heads, n_points, ratio, cffn_ratio, with_cffn) for _ in range(2)]) if extra else None

    def forward(self, x, c, blocks, d1, d2, H, W):
        x = self.injector(x, d1[0], c, d1[1], d1[2])
        for blk in blocks:
            x = blk(x, H, W)
        c = self.extractor(c, d2[0], x, d2[1], d2[2], H, W)
        if self.extra_extractors is not None:
            for e in self.extra_extractors:
                c = e(c, d2[0], x, d2[1], d2[2], H, W)
        return x, c


# ----------------------------------------------------------------------------- backbone (BK)
class OracleEncoder(nn.Module):
    """Restates BK:27-349 (`SAMAdapterbimodalMixModNewInTwinConvNEW`), eval-mode semantics."""

    def __init__(self, img_size=1024, patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0,
                 window_size=14, global_attn_indexes=(5, 11, 17, 23), pretrained_size=1024,
                 conv_inplane=48, n_points=4, deform_num_heads=16, init_values=1e-6, interaction_indexes=None,
                 cffn_ratio=0.25, deform_ratio=0.5, arch="small", with_cffn=True, use_extra_extractor=True, add_vit_feature=True,
                 qkv_bias=True, use_rel_pos=True, **_ignored):
        super().__init__()
        self.img_size, self.embed_dim = img_size, embed_dim
        self.add_vit_feature = add_vit_feature   # BK:33,44,326
        self.interaction_indexes = interaction_indexes
        grid = pretrained_size // patch_size
        # registration order follows IE:235-276 then BK:54-99 so that state_dict ordering matches
        self.patch_embed = PatchEmbed(patch_size, 3, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, grid, grid, embed_dim))
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, window_size if i not in global_attn_indexes else 0, (grid, grid), qkv_bias, use_rel_pos)
            for i in range(depth)])
        self.spm = SpatialPriorModuleBimodal(conv_inplane, embed_dim, img_size, arch)
        self.up = nn.ConvTranspose2d(embed_dim, embed_dim, 2, 2)
        self.level_embed = nn.Parameter(torch.zeros(3, embed_dim))
        n = len(interaction_indexes)
        self.interactions = nn.Sequential(*[
            InteractionBlock(embed_dim, deform_num_heads, n_points, deform_ratio, cffn_ratio, init_values,
                             i == n - 1 and use_extra_extractor, with_cffn)   # BK:86-94
            for i in range(n)])
        self.norm1 = nn.BatchNorm2d(embed_dim)
        self.norm2 = nn.BatchNorm2d(embed_dim)
        self.norm3 = nn.BatchNorm2d(embed_dim)
        self.norm4 = nn.BatchNorm2d(embed_dim)
        nn.init.normal_(self.level_embed)
        self.eval()

    @torch.no_grad()
    def forward(self, x, taps=None):
        x_other = x[:, 3:]
        x = x[:, :3]
        c1, c2, c3, c4 = self.spm(x, x_other, taps)
        c2 = c2 + self.level_embed[0]
        c3 = c3 + self.level_embed[1]
        c4 = c4 + self.level_embed[2]
        c = torch.cat([c2, c3, c4], dim=1)
        if taps is not None:
            taps["c1_map"], taps["c_in"] = c1, c   # (c1_map: "c1" is the adapter tokens after interaction 1 below)
        d1, d2 = deform_inputs(x.shape[2], x.shape[3], x.dtype)
        x, H, W = self.patch_embed(x)
        bs, n, dim = x.shape
        pe = F.interpolate(self.pos_embed.permute(0, 3, 1, 2), size=(H, W), mode="bicubic",
                           align_corners=False).reshape(1, -1, H * W).permute(0, 2, 1)
        x = x + pe
        if taps is not None:
            taps["x_in"] = x
        outs = []
        for i, layer in enumerate(self.interactions):
            idx = self.interaction_indexes[i]
            x, c = layer(x, c, self.blocks[idx[0]:idx[-1] + 1], d1, d2, H, W)
            if taps is not None:
                taps[f"x{i}"], taps[f"c{i}"] = x, c
            outs.append(x.transpose(1, 2).view(bs, dim, H, W).contiguous())
        c1 = c1.transpose(1, 2).view(bs, dim, 4 * H, 4 * W).contiguous()
        n2, n3 = c2.size(1), c3.size(1)
        c2 = c[:, :n2].transpose(1, 2).view(bs, dim, H * 2, W * 2).contiguous()
        c3 = c[:, n2:n2 + n3].transpose(1, 2).view(bs, dim, H, W).contiguous()
        c4 = c[:, n2 + n3:].transpose(1, 2).view(bs, dim, H // 2, W // 2).contiguous()
        c1 = self.up(c2) + c1
        if self.add_vit_feature:   # BK:326-331
            x1 = F.interpolate(outs[0], scale_factor=4, mode="bilinear", align_corners=False)
            x2 = F.interpolate(outs[1], scale_factor=2, mode="bilinear", align_corners=False)
            x4 = F.interpolate(outs[3], scale_factor=0.5, mode="bilinear", align_corners=False)
            c1, c2, c3, c4 = c1 + x1, c2 + x2, c3 + outs[2], c4 + x4
        return [self.norm1(c1), self.norm2(c2), self.norm3(c3), self.norm4(c4)], None
