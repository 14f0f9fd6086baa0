"""MI355X drop-in for the reference's `SegformerHead` (segmentation/mmseg_custom/models/decode_heads/segformer_head.py:11-66;
decode_head dict of configs/DELIVER/Segformer_MMSAM_adapter_large_DELIVER_1024x1024_ss_RGBLIDAR_hard.py:57-66).

Same constructor kwargs, same `state_dict()` keys/shapes/order (`conv_seg.*`, `convs.i.{conv,bn}.*`,
`fusion_conv.{conv,bn}.*`), `forward(list of 4 NCHW maps) -> logits [B, num_classes, H/4, W/4]` fp32 -- the tensor the
data-parallel harness all-gathers.  Inference only.

Device schedule (all through the C ABI, include/mmsa.h):
  per branch i : nchw_to_planes(f_i) -> GEMM (W_i with the BN scale folded in, + BN shift, ReLU) -> planes y_i
                 -> GEMM y_i x Wfuse[:, i-th slice] -> z_i fp32 at the branch's own resolution
  head_fuse    : z_0 + sum_i bilinear(z_i) -> fusion BN -> ReLU -> planes            (csrc/head.hip: why this is exact algebra)
  classifier   : GEMM (+bias) -> tokens_to_nchw
"""
import torch
import torch.nn as nn

from . import lib
from . import ops
from .params import Node, _attach


class SegformerHead(nn.Module):
    def __init__(self, interpolate_mode="bilinear", in_channels=None, channels=None, *, num_classes=None, dropout_ratio=0.1,
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), in_index=-1, input_transform="multiple_select",
                 loss_decode=None, ignore_index=255, sampler=None, align_corners=False, init_cfg=None, **kwargs):
        super().__init__()
        if in_channels is None or channels is None or num_classes is None:
            raise TypeError("SegformerHead: in_channels, channels and num_classes are required")
        if not isinstance(in_channels, (list, tuple)) or not isinstance(in_index, (list, tuple)) or len(in_channels) != len(in_index):
            raise AssertionError("SegformerHead: in_channels and in_index must be lists of equal length")  # segformer_head.py:30
        if not 1 <= len(in_channels) <= 4:
            raise NotImplementedError("mmsa SegformerHead: 1..4 input maps")
        if interpolate_mode != "bilinear" or align_corners:
            raise NotImplementedError("mmsa SegformerHead: only interpolate_mode='bilinear', align_corners=False (the reference configs)")
        if norm_cfg is None or norm_cfg.get("type") not in ("BN", "SyncBN"):
            raise NotImplementedError("mmsa SegformerHead: norm_cfg must be BN / SyncBN (eval: running-statistics affine)")
        if act_cfg is None or act_cfg.get("type") != "ReLU":
            raise NotImplementedError("mmsa SegformerHead: act_cfg must be ReLU")
        if channels % 8:
            raise NotImplementedError("mmsa SegformerHead: channels must be a multiple of 8")
        self.in_channels, self.in_index, self.channels = list(in_channels), list(in_index), channels
        self.num_classes, self.dropout_ratio, self.align_corners = num_classes, dropout_ratio, align_corners
        self.interpolate_mode, self.ignore_index, self.norm_cfg, self.act_cfg = interpolate_mode, ignore_index, norm_cfg, act_cfg
        n = len(in_channels)
        g = torch.Generator().manual_seed(0)

        def conv_w(co, ci):  # kaiming-normal like mmcv ConvModule.init_weights
            return torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5

        def bn(prefix):
            _attach(self, prefix + ".weight", torch.ones(channels))
            _attach(self, prefix + ".bias", torch.zeros(channels))
            _attach(self, prefix + ".running_mean", torch.zeros(channels), buffer=True)
            _attach(self, prefix + ".running_var", torch.ones(channels), buffer=True)
            _attach(self, prefix + ".num_batches_tracked", torch.zeros((), dtype=torch.long), buffer=True)

        _attach(self, "conv_seg.weight", torch.randn(num_classes, channels, 1, 1, generator=g) * 0.01)  # normal_init std 0.01
        _attach(self, "conv_seg.bias", torch.zeros(num_classes))
        self.add_module("convs", Node())
        for i, ci in enumerate(in_channels):
            _attach(self, f"convs.{i}.conv.weight", conv_w(channels, ci))
            bn(f"convs.{i}.bn")
        _attach(self, "fusion_conv.conv.weight", conv_w(channels, channels * n))
        bn("fusion_conv.bn")
        self._packed = None
        self._bufs = {}
        self.register_load_state_dict_post_hook(lambda m, _: m.invalidate())
        self.eval()

    def invalidate(self):
        self._packed = None

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("mmsa: the MI355X decode head implements the inference forward path only")
        return super().train(False)

    def init_weights(self):
        pass

    def forward_test(self, inputs, img_metas=None, test_cfg=None):
        """BaseDecodeHead.forward_test (mmseg 0.20.2; called by EncoderDecoder._decode_head_forward_test, ED:129-133)."""
        return self.forward(inputs)

    def forward_train(self, inputs, img_metas, gt_semantic_seg, train_cfg):
        raise NotImplementedError("mmsa: the MI355X decode head implements the inference forward path only (no losses)")

    @torch.no_grad()
    def _pack(self, dev):
        sd = {k: v.detach().to(dev, torch.float32) for k, v in self.state_dict().items() if v.dtype.is_floating_point}
        ch, n = self.channels, len(self.in_channels)

        def bn_affine(p):
            s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + 1e-5)
            return s.contiguous(), (sd[p + ".bias"] - sd[p + ".running_mean"] * s).contiguous()

        pk = {"branch": []}
        wf = sd["fusion_conv.conv.weight"].reshape(ch, ch * n)
        for i, ci in enumerate(self.in_channels):
            s, t = bn_affine(f"convs.{i}.bn")
            w = sd[f"convs.{i}.conv.weight"].reshape(ch, ci) * s[:, None]
            pk["branch"].append(dict(w=ops.split_planes(w.contiguous()), shift=t,
                                     wf=ops.split_planes(wf[:, ch * i:ch * (i + 1)].contiguous())))
        pk["fs"], pk["ft"] = bn_affine("fusion_conv.bn")
        ncp = ops.pad32(self.num_classes)
        wc = torch.zeros(ncp, ch, device=dev)
        wc[:self.num_classes] = sd["conv_seg.weight"].reshape(self.num_classes, ch)
        bc = torch.zeros(ncp, device=dev)
        bc[:self.num_classes] = sd["conv_seg.bias"]
        pk["wc"], pk["bc"], pk["ncp"] = ops.split_planes(wc), bc, ncp
        return pk

    def _buf(self, name, shape, dtype=torch.float32, dev=None):
        name = getattr(self, "buf_tag", "") + name     # mmsa.chains: one set of scratch buffers per concurrent chain
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
            t = torch.zeros(shape, dtype=dtype, device=dev)
            self._bufs[name] = t
        return t

    def _planes(self, name, rows, cols, dev):
        kp = ops.pad32(cols)
        return ops.Planes(self._buf(name, (rows, 2 * kp), torch.int16, dev), rows, cols, kp)

    @torch.no_grad()
    def forward(self, inputs, out=None):
        """out: optional preallocated fp32 [B, num_classes, H/4, W/4] GPU tensor to write the logits into (mmsa.Chains gives every chain its slice
        of the step's logits); default: a fresh tensor per call, like the reference's."""
        x0 = inputs[self.in_index[0]]
        if not x0.is_cuda:
            raise RuntimeError("mmsa SegformerHead: inputs must live on the GPU (there is no CPU path)")
        with torch.cuda.device(x0.device):    # launches go to the current stream of the tensors' device
            return self._forward(inputs, out)

    def _forward(self, inputs, out=None):
        xs = [inputs[i] for i in self.in_index]          # BaseDecodeHead._transform_inputs('multiple_select')
        x0 = xs[0]
        dev = x0.device
        if self._packed is None or self._packed["wc"].p.device != dev:
            self._packed = self._pack(dev)
        pk = self._packed
        B, _, H, W = x0.shape
        ch = self.channels
        zs = []
        for i, (x, ci) in enumerate(zip(xs, self.in_channels)):
            if x.dtype != torch.float32 or x.dim() != 4 or x.shape[0] != B or x.shape[1] != ci:
                raise RuntimeError(f"mmsa SegformerHead: input {i} must be fp32 [B={B}, {ci}, h, w], got {tuple(x.shape)} {x.dtype}")
            x = x.contiguous()
            h, w = x.shape[2:]
            rows = B * h * w
            # planes of this map written by the backbone's tail (emit_planes).  They live in the backbone's workspace and are
            # overwritten by its NEXT forward: `live()` is false for the maps of an earlier call, which are re-derived from the
            # NCHW tensor instead of silently reading the other image's planes.
            xp = getattr(inputs[self.in_index[i]], "_mmsa_planes", None)
            if not isinstance(xp, ops.Planes) or xp.n != rows or xp.k != ci or xp.p.device != dev or not xp.live():
                xp = self._planes(f"x{i}", rows, ci, dev)
                lib.call("mmsa_nchw_to_planes", x.data_ptr(), ci * h * w, xp.p.data_ptr(), 2 * xp.kpad, B, ci, h * w, ops._stream())
            br = pk["branch"][i]
            yp = self._planes(f"y{i}", rows, ch, dev)
            ops.gemm(xp, br["w"], bias=br["shift"], act="relu", out_planes=yp)
            z = self._buf(f"z{i}", (rows, ch), dev=dev)
            ops.gemm(yp, br["wf"], z)
            zs.append((z, h, w))
        fp = self._planes("fused", B * H * W, ch, dev)
        lv = zs[1:] + [(None, 0, 0)] * (4 - len(zs))
        lib.call("mmsa_head_fuse", zs[0][0].data_ptr(),
                 lv[0][0].data_ptr() if lv[0][0] is not None else None, lv[0][1], lv[0][2],
                 lv[1][0].data_ptr() if lv[1][0] is not None else None, lv[1][1], lv[1][2],
                 lv[2][0].data_ptr() if lv[2][0] is not None else None, lv[2][1], lv[2][2],
                 ch, pk["fs"].data_ptr(), pk["ft"].data_ptr(), fp.p.data_ptr(), 2 * fp.kpad, None, 0, B, H, W, ch, ops.ACT["relu"],
                 ops._stream())
        lt = self._buf("logit_tokens", (B * H * W, pk["ncp"]), dev=dev)
        ops.gemm(fp, pk["wc"], lt, bias=pk["bc"])
        if out is None:
            out = torch.empty(B, self.num_classes, H, W, device=dev)   # a fresh tensor per call, like the reference's (no aliasing across calls)
        elif (tuple(out.shape) != (B, self.num_classes, H, W) or out.dtype != torch.float32 or out.device != dev or not out.is_contiguous()):
            raise RuntimeError(f"mmsa SegformerHead: out must be a contiguous fp32 [{B}, {self.num_classes}, {H}, {W}] tensor on {dev}")
        lib.call("mmsa_tokens_to_nchw", lt.data_ptr(), pk["ncp"], out.data_ptr(), B, H * W, self.num_classes, ops._stream())
        return out
