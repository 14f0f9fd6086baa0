"""Drop-in backbone `SAMAdapterbimodalMixModNewInTwinConvNEW` (BK:27-349) running on the MI355X HIP library.

Same class name, constructor kwargs (CFG-L:30-56), `forward(x[B,6,H,W]) -> ([f1,f2,f3,f4] NCHW fp32, None)`,
`init_weights(pretrained)` and state_dict keys as the reference, so mmseg's EncoderDecoder, the Segformer head,
the configs and the checkpoint loader see the same plugin.  Inference (eval) semantics only: DropPath / Dropout /
activation checkpointing are identities, SyncBatchNorm uses running statistics.

All arithmetic is done by hand-written HIP kernels through the C ABI (mmsa.ops); torch allocates device buffers
and provides the stream.  Activations are fp32, token-major (NHWC) end to end; only the four outputs are written
NCHW by the fused tail kernel.  There is no CPU / PyTorch fallback."""
import contextlib
import math

import os

import torch
import torch.nn as nn

from . import ops
from .params import CONVNEXT_ARCH, build_tree

_VIT_DEFAULTS = dict(img_size=1024, patch_size=16, in_chans=3, embed_dim=1024, depth=24, mlp_ratio=4.0,
                     qkv_bias=True, use_abs_pos=True, use_rel_pos=True, rel_pos_zero_init=True, window_size=14,
                     global_attn_indexes=[5, 11, 17, 23], pretrained_size=1024, fix=False)


class OperandRangeError(RuntimeError):
    """A forward converted a value beyond the range of an fp16-based operand format (the clamp watch word, include/mmsa.h) and nothing is left to
    re-route: `range_fallback` is off, or the value is a q / k / v (or a qkv bias / rel-pos table entry) beyond +-65504 inside the attention kernels,
    which read fp16-based planes in every mode.  The outputs of that forward are invalid."""


class Workspace:
    """Named, grow-only device buffers (stable addresses once warmed up -> HIP-graph capturable)."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}
        self.zeroed = set()   # buffers whose never-written parts must stay zero (K padding of planes)

    def get(self, name, rows, cols, dtype=torch.float32, zero=False):
        n = rows * cols
        b = self.bufs.get(name)
        if b is None or b.numel() < n or b.dtype != dtype:
            b = torch.zeros(n, dtype=dtype, device=self.device) if zero else torch.empty(n, dtype=dtype, device=self.device)
            self.bufs[name] = b
        if zero:
            self.zeroed.add(name)
        return b[:n].view(rows, cols)

    def poison(self):
        """Testing aid: fill every scratch buffer that is not of the keep-zero kind with NaN (fp) / 0x7fc0 bf16 NaN patterns,
        so that a read of a never-written element shows up in the outputs (tests/test_backbone_gpu.py)."""
        for name, b in self.bufs.items():
            if name in self.zeroed:
                continue
            if b.dtype in (torch.float32, torch.float64):
                b.fill_(float("nan"))
            elif b.dtype == torch.int16:
                b.fill_(0x7fc0)

    def planes(self, name, rows, cols, zero=False, fmt=ops.FMT_B3):
        """Activation planes of a [rows, cols] matrix (cols padded to 32) in the given operand format (ops.Planes)."""
        tr, tc, cp = ops.planes_shape(rows, cols, fmt)
        return ops.Planes(self.get(name + ".pl", tr, tc, torch.int16, zero or cp != cols), rows, cols, cp, fmt)

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self.bufs.values())


class _Tagged:
    """Workspace view that prefixes buffer names (private buffers for a chain that runs on its own stream)."""

    def __init__(self, ws, tag):
        self.ws, self.tag = ws, tag

    def get(self, name, *a, **k):
        return self.ws.get(self.tag + name, *a, **k)

    def planes(self, name, *a, **k):
        return self.ws.planes(self.tag + name, *a, **k)

    @property
    def device(self):
        return self.ws.device

    def poison(self):
        return self.ws.poison()

    def nbytes(self):
        return self.ws.nbytes()


class SAMAdapterbimodalMixModNewInTwinConvNEW(nn.Module):
    def __init__(self, pretrain_size=1024, num_heads=12, conv_inplane=64, n_points=4,
                 modalities_name=['rgb', 'depth', 'lidar', 'event'], modalities_ch=[3, 3, 3, 1],
                 deform_num_heads=6, init_values=0., gamma_init_values=0., interaction_indexes=None, with_cffn=True,
                 cffn_ratio=0.25, deform_ratio=1.0, add_vit_feature=True, pretrained=None,
                 use_extra_extractor=True, with_cp=True, drop_path_rate=0.4, drop_rate=0., drop_multimodal_path=0.2,
                 arch='base', checkpoint='check', *args, **kwargs):
        super().__init__()
        vit = dict(_VIT_DEFAULTS)
        for k in list(kwargs):
            if k in vit:
                vit[k] = kwargs.pop(k)
        # remaining kwargs (e.g. conv_drop_path_rate of the *withcp variant, norm_layer, act_layer) are
        # training-only or fixed by the architecture and are accepted like BK:34 does with *args/**kwargs.
        if 'rgb' not in modalities_name or len(modalities_name) != 2:
            raise NotImplementedError("mmsa: the MI355X path implements the bimodal (rgb + one auxiliary modality) encoder; "
                                      f"got modalities_name={modalities_name}")
        if list(modalities_ch) != [3, 3]:
            raise NotImplementedError("mmsa: TwinConvNeXt takes two 3-channel streams (TC:296-316)")
        if not vit["use_abs_pos"]:
            raise NotImplementedError("mmsa: use_abs_pos=False is not a working configuration of the reference either (its forward interpolates "
                                      "self.pos_embed unconditionally, BK:276, which is None then: IE:113-114)")
        if interaction_indexes is None:
            raise ValueError("interaction_indexes is required")
        D = vit["embed_dim"]
        self.cfg = dict(embed_dim=D, depth=vit["depth"], num_heads=num_heads, mlp_ratio=vit["mlp_ratio"],
                        patch_size=vit["patch_size"], pretrained_size=vit["pretrained_size"], img_size=vit["img_size"],
                        window_size=vit["window_size"], global_attn_indexes=list(vit["global_attn_indexes"]),
                        conv_inplane=conv_inplane, n_points=n_points, deform_num_heads=deform_num_heads,
                        init_values=init_values, interaction_indexes=[list(i) for i in interaction_indexes],
                        cffn_ratio=cffn_ratio, deform_ratio=deform_ratio, arch=arch,
                        use_extra_extractor=bool(use_extra_extractor), with_cffn=bool(with_cffn), add_vit_feature=bool(add_vit_feature),
                        use_rel_pos=bool(vit["use_rel_pos"]), qkv_bias=bool(vit["qkv_bias"]))
        a = CONVNEXT_ARCH[arch] if isinstance(arch, str) else arch
        self.depths, self.channels = list(a["depths"]), list(a["channels"])
        if [2 * c for c in self.channels] != [conv_inplane * m for m in (4, 8, 16, 32)]:
            raise ValueError("conv_inplane*{4,8,16,32} must equal twice the ConvNeXt stage widths (AM:894-907)")
        for c in self.channels:
            if c % 32 != 0:
                raise ValueError("ConvNeXt stage widths must be multiples of 32 (groups=32 convs, GroupNorm(32))")
        hd = D // num_heads
        if D % num_heads or ops.pad32(hd) not in (32, 64, 96):
            raise NotImplementedError(f"mmsa: attention head_dim {hd} not supported by the HIP kernels (up to 96; widths that are "
                                      "not multiples of 32 run zero-padded per head, e.g. ViT-H's 80 as 96)")
        self.img_size = vit["img_size"]
        self.embed_dim = D
        self.interaction_indexes = self.cfg["interaction_indexes"]
        self.modalities_name, self.modalities_ch = modalities_name, modalities_ch
        self.in_ch_im = 3
        build_tree(self, self.cfg)
        self._packed = None
        self._ws = None
        self._taps = None     # per-stage copies for the error-budget test (forward_taps); None on the product path
        self.register_load_state_dict_post_hook(lambda m, _: m.invalidate())
        self.init_weights(pretrained)
        # TwinConvNeXt.init_weights (TC:382-443) loads `checkpoint` -- a URL in every shipped config.  There is no network here:
        # a local mirror can be named in MMSA_CONVNEXT_CKPT; a checkpoint that cannot be read is reported, never skipped silently.
        if isinstance(checkpoint, str) and checkpoint not in ("", "check"):
            local = checkpoint if os.path.isfile(checkpoint) else os.environ.get("MMSA_CONVNEXT_CKPT", "")
            if os.path.isfile(local):
                self.load_convnext_checkpoint(local)
            else:
                import warnings
                warnings.warn(f"mmsa: ConvNeXt checkpoint '{checkpoint}' is not a local file and MMSA_CONVNEXT_CKPT names no mirror: "
                              "TwinConvNeXt keeps its random initialisation (the reference downloads and loads it, TC:382-443)")
        self.eval()

    # ------------------------------------------------------------------ plugin surface
    def init_weights(self, pretrained=None):
        """IE:305-315: a path loads a (SAM) checkpoint with the reference's non-strict loader semantics
        (mmcv_custom/checkpoint.py:319-514, restated in mmsa/checkpoint.py: wrappers, 'module.' / 'encoder.' prefixes, size-mismatched
        keys skipped with a warning)."""
        if isinstance(pretrained, str):
            from .checkpoint import load_pretrained
            self._pretrained_report = load_pretrained(self, pretrained)

    def load_convnext_checkpoint(self, path):
        """TwinConvNeXt.init_weights (TC:403-443): a single-stream ConvNeXt checkpoint is loaded into BOTH streams -- every key
        gets '_x' / '_y' inserted before its first '.' ('stages.0.0.gamma' -> 'stages_x.0.0.gamma' and 'stages_y...') and is
        loaded non-strictly, so keys that do not exist under the new name (the reference's 'norm0.weight' -> 'norm0_x.weight',
        while its modules are called norm_x0) are skipped exactly like there.  Returns the list of twin_conv keys loaded."""
        ck = torch.load(path, map_location="cpu")
        sd = ck.get("state_dict", ck.get("model", ck)) if isinstance(ck, dict) else ck
        sd = {(k[9:] if k.startswith("backbone.") else k): v for k, v in sd.items()}
        if list(sd.keys())[0].startswith("module."):
            sd = {k[7:]: v for k, v in sd.items()}
        own = self.state_dict()
        loaded, new = [], {}
        for k, v in sd.items():
            dot = k.find(".")
            for sfx in ("_x", "_y"):
                nk = (k[:dot] + sfx + k[dot:]) if dot != -1 else k + sfx
                full = "spm.twin_conv." + nk
                if full in own:   # a shape mismatch raises in load_state_dict, like in the reference
                    new[full] = v
                    loaded.append(full)
        self.load_state_dict(new, strict=False)
        return loaded

    def invalidate(self):
        """Drop packed weights (call after mutating parameters in place).  New weights also start outside the wide-range state (`range_fallback`): whether
        they need it is measured again by the clamp watch of the next pack / forward."""
        self._packed = None
        self._wide_range = False
        self._carry_modes = None
        self._inter_pairs = set()

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("mmsa: the MI355X backbone implements the inference forward path only")
        return super().train(False)

    # GEMM sites whose operands travel in the h8 format (ops.Planes: fp16 hi + e5m2 cross-term bytes, 2/3 of the matrix-pipe time
    # of the bf16 hi/lo scheme; csrc/common.h has the error analysis, tools/precision_study.py the end-to-end measurement).
    # "vit" = qkv / proj / lin1 / lin2 of the SAM ViT blocks; "inter" = the Linear layers of the injectors / extractors (MSDA
    # projections, ConvFFN); "up" = the 2x2 transposed conv of the tail.  The TwinConvNeXt / neck GEMMs stay on bf16 hi/lo (the most
    # error-sensitive part of the path: SURVEY appendix F).  `model.h8_sites = ()` keeps every site on 16-bit hi/lo pairs.
    # Every switch of this class that changes numerics is an ATTRIBUTE (h8_sites, h8c, cnx_f16, fold_ln, fold_adapter_ln, fold_convnext_ln, share_c_norm,
    # fuse_convnext_mlp, attention_precision, attention_guard, range_fallback, inter_follow_blocks, msda_value): set it before the first forward, or call invalidate() after changing it; the A/B tools pass
    # them through `bench.py --set attr=value`.  No environment variable changes what this module computes.
    # "attnv" = the attention kernels run P V on the fp16 MFMA: the v third of the qkv planes (GEMM output and bias rows) is h8-encoded
    # and P is rounded to fp16 (csrc/attention.hip VF; 2.9e-5 on the ViT-B oracle study, LAB_NOTES.md 4.1); the kernels with the rel-pos terms
    # fused run Q K^T and the rel-pos terms on fp16 hi parts too -- per block, and only while its logits are small (_attn_mode).
    H8_DEFAULT = ("vit", "inter", "up", "attnv")
    attention_precision = "auto"   # 'auto' | 'f16' | 'b3': operand precision INSIDE the attention kernels where the "attnv" site allows fp16 (_attn_mode)

    # Range fallback (round 6; VERDICT r05 weak 1: "an input the reference computes can be refused").  The fp16-based operand formats clamp at +-57344
    # (h8 / h8c) / +-65504 (f3); the reference is fp32 and has no such limit.  A forward -- or the pack itself -- that had to clamp (the clamp watch word)
    # switches the MODEL to its wide-range state: every GEMM site that travels on fp16-based planes (ViT blocks, interactions, up-conv, TwinConvNeXt) is
    # repacked on bf16 hi/lo pairs, which have fp32's exponent range (2^-17 per product instead of 2^-15.6 / 2^-22: the formats of rounds 1-3, error
    # budget in LAB_NOTES.md section 2), and the forward runs again -- exactly the contract of the attention logit guard: eager forwards re-route themselves,
    # graph owners see `check_attention_guard()` report every block as moved and capture again.  The state is sticky (it travels with the packed file) and
    # one-way.  What stays fp16-based in every mode: the planes the attention kernels read (q, k, v, bias rows, rel-pos tables; |x| <= 65504 / 57344) --
    # a q . k with such entries is a logit beyond 1e9, where fp32 softmax itself is a one-hot of rounding noise; that case still raises OperandRangeError.
    # `model.range_fallback = False`: refuse instead of switching (the behaviour of round 5).
    range_fallback = True

    # Interaction sites follow their blocks (round 6; VERDICT r05 weak 2, LAB_NOTES 6 "next as of round 4 (8)").  When the logit guard moves a ViT block off
    # single fp16 operands, what then limits a checkpoint with peaky attention is no longer the block but what the h8c injector / extractor GEMMs of ITS
    # interaction feed into its q and k (LAB_NOTES 2: at max |logit| 32 in all 24 blocks the probes are 0.8-2.0e-4 with the default sites, 0.5-1.0e-4 with the
    # interaction sites on pairs too).  So an interaction whose group holds a moved block is repacked on the pair format as well (`_inter_pairs`: a set of
    # interaction indices, sticky like the blocks' modes, stored with the packed file).  `inter_follow_blocks = False`: the behaviour of rounds 3-5.
    inter_follow_blocks = True

    def _inter_pair_set(self):
        return set(getattr(self, "_inter_pairs", ()) or ())

    def _interaction_of_block(self, bi):
        for i, idx in enumerate(self.interaction_indexes):
            if idx[0] <= bi <= idx[-1]:
                return i
        return None

    def _wide(self):
        return bool(getattr(self, "_wide_range", False))

    def _pair_fmt(self):
        """The hi/lo PAIR format of the ViT-block GEMMs (h8 off, or a block moved off single fp16 operands by the logit guard): fp16 pairs (f3: 22
        significant bits, +-65504), or bf16 pairs (16 bits, fp32's range) in the wide-range state."""
        return ops.FMT_B3 if self._wide() else ops.FMT_F3

    def _h8_sites(self):
        sites = tuple(getattr(self, "h8_sites", self.H8_DEFAULT))
        if self._wide():   # wide-range state: no GEMM site on fp16-based planes ("attnv", the attention kernels' own operands, stays)
            sites = tuple(s_ for s_ in sites if s_ not in ("vit", "inter", "up", "cnx", "cnx2"))
        return sites

    @contextlib.contextmanager
    def chain(self, index):
        """Everything forward() does inside this context uses chain `index`'s private scratch buffers (a tagged view of the
        workspace), so that several sub-batches can be in flight on concurrent streams with ONE set of packed weights
        (mmsa.chains).  Results are those of a plain forward()."""
        base = self._ws
        if base is None:
            raise RuntimeError("mmsa: run one forward() before opening a chain (workspace not created yet)")
        self._ws = _Tagged(base, f"chain{index}_")
        try:
            yield self
        finally:
            self._ws = base

    # ------------------------------------------------------------------ packing (one-time weight preprocessing)
    @torch.no_grad()
    def _block_gemm_planes(self, sd, i, fmt, fold, dev):
        """The four GEMM weights of ViT block i as operand planes of format `fmt` (+ the LayerNorm-folded forms of qkv / lin1 when
        `fold`): head widths that are not a multiple of 32 zero-padded per head (ViT-H), see _pack."""
        heads_, hd_true, hd_ = self.cfg["num_heads"], self._hd_true, self._hd_pad

        def pad_head_rows(w, groups):
            if hd_ == hd_true:
                return w.contiguous()
            v = w.reshape(groups, heads_, hd_true, *w.shape[1:])
            out = v.new_zeros(groups, heads_, hd_, *w.shape[1:])
            out[:, :, :hd_true] = v
            return out.reshape(groups * heads_ * hd_, *w.shape[1:]).contiguous()

        def planes(w2d):
            return ops.split_planes(w2d.contiguous(), None, fmt=fmt, weight=fmt == ops.FMT_H8)

        def folded(w, bias, lnw, lnb):
            """(planes of W o lnw, column sums of those planes as the kernel will read them, W lnb + bias)"""
            pl = planes(w * lnw[None, :])
            cs = ops.planes_to_float(pl, cols=w.shape[1])[: w.shape[0]].double().sum(1).float().contiguous()
            return pl, cs, (w.double() @ lnb.double()).float().add_(bias).contiguous()
        b = f"blocks.{i}."
        g = lambda k: sd[b + k].detach().to(dev, torch.float32)   # noqa: E731
        qkv_w, qkv_bias = pad_head_rows(g("attn.qkv.weight"), 3), pad_head_rows(g("attn.qkv.bias"), 3)
        proj_w = pad_head_rows(g("attn.proj.weight").t(), 1).t().contiguous()    # zero COLUMNS for the pad channels
        out = dict(qkv=planes(qkv_w), proj=planes(proj_w), lin1=planes(g("mlp.lin1.weight")), lin2=planes(g("mlp.lin2.weight")))
        if fold:   # the consumers' weights carry their LayerNorm (the unfolded planes are not kept: the fold is all or nothing per model)
            out["qkv"], out["qkv_cs"], out["qkv_bf"] = folded(qkv_w, qkv_bias, g("norm1.weight"), g("norm1.bias"))
            out["lin1"], out["lin1_cs"], out["lin1_bf"] = folded(g("mlp.lin1.weight"), g("mlp.lin1.bias"), g("norm2.weight"), g("norm2.bias"))
        return out

    def _pack_state_dict(self, dev):
        """The float parameters on `dev`, plus ZERO stand-ins for the tensors a ViT built with qkv_bias=False / use_rel_pos=False does not
        have (IE:317,320-327): a zero qkv bias and zero rel-pos tables make the kernels compute exactly what the reference computes
        without them (the pad tokens' k / v = bias = 0; rel_pos . q = 0 added to the logits)."""
        sd = {k: v.detach().to(dev, torch.float32) for k, v in self.state_dict().items() if v.dtype.is_floating_point}
        cfg = self.cfg
        D, hd = cfg["embed_dim"], cfg["embed_dim"] // cfg["num_heads"]
        grid = cfg["pretrained_size"] // cfg["patch_size"]
        for i in range(cfg["depth"]):
            b = f"blocks.{i}.attn."
            if not cfg["qkv_bias"]:
                sd[b + "qkv.bias"] = torch.zeros(3 * D, device=dev)
            if not cfg["use_rel_pos"]:
                L = 2 * (grid if i in cfg["global_attn_indexes"] else cfg["window_size"]) - 1
                sd[b + "rel_pos_h"] = torch.zeros(L, hd, device=dev)
                sd[b + "rel_pos_w"] = torch.zeros(L, hd, device=dev)
        return sd

    def _h8c_wanted(self):
        """h8 sites with deep contractions travel as h8c planes (default); `model.h8c = False` keeps them on h8 line planes (A/B)."""
        return bool(getattr(self, "h8c", True))

    def _cnx_f16_wanted(self):
        """TwinConvNeXt GEMMs on fp16 hi/lo pairs (default) instead of bf16 hi/lo; `model.cnx_f16 = False` for the A/B.  Not with
        the opt-in ConvNeXt LayerNorm fold (its depthwise kernel writes bf16 hi/lo planes)."""
        fold = bool(getattr(self, "fold_convnext_ln", False))
        return bool(getattr(self, "cnx_f16", True)) and not fold and not self._wide()

    def _fold_ln_wanted(self, hidden=None):
        """Whether _pack folds the ViT blocks' LayerNorms into their consumer GEMMs (a pack-time setting: checkpoint.load_packed compares it)."""
        D, heads = self.cfg["embed_dim"], self.cfg["num_heads"]
        Da = heads * ops.pad32(D // heads)
        hidden = int(D * self.cfg["mlp_ratio"]) if hidden is None else hidden
        return (bool(getattr(self, "fold_ln", True)) and D % 64 == 0
                and (3 * Da) % 128 == 0 and hidden % 128 == 0)

    def _fold_adapter_ln_wanted(self):
        """Whether _pack folds the adapter tokens' LayerNorms into their consumer GEMMs (opt-in `fold_adapter_ln`; a pack-time setting that drives the
        run-time path: checkpoint.load_packed compares it)."""
        cfg = self.cfg
        D = cfg["embed_dim"]
        hid_c = int(D * cfg["cffn_ratio"]) if cfg["with_cffn"] else 128
        return bool(getattr(self, "share_c_norm", True) and getattr(self, "fold_adapter_ln", False) and "inter" in self._h8_sites() and D % 64 == 0
                    and int(D * cfg["deform_ratio"]) % 128 == 0 and hid_c % 128 == 0
                    and not self._inter_pair_set())   # (the fold hands c's planes from one interaction's producer to the next one's consumers: one format for all)

    @torch.no_grad()
    def _pack(self, dev):
        """Pre-pack the weights for `dev`.  The guard words of the model live here too: one logit word per ViT block (check_attention_guard) and, last, the
        clamp watch word (include/mmsa.h "Clamp watch") -- weight planes that had to clamp a value report into it like every forward's activations do."""
        depth = self.cfg["depth"]
        guard = torch.zeros(depth + 1, device=dev)
        carry, self._carry_modes = getattr(self, "_carry_modes", None), None
        with ops.clamp_watch(guard[depth:]):
            pk = self._pack_impl(dev)
            if carry is not None and len(carry) == len(pk["blocks"]):
                # a pack was dropped whose blocks had settled their attention precision (the switch to the wide-range state; interactions following their
                # blocks): the decisions carry over, and a block on pair attention runs its four GEMMs on pairs too (check_attention_guard)
                sd_dev = None
                for bp, (amode, ml) in zip(pk["blocks"], carry):
                    bp["max_logit"] = ml
                    if amode is not None:
                        bp["amode"] = amode
                    if amode == "b3" and bp["qkv"].fmt != self._pair_fmt():
                        if sd_dev is None:
                            sd_dev = self._pack_state_dict(dev)
                        bp.update(self._block_gemm_planes(sd_dev, bp["index"], self._pair_fmt(), pk["fold_ln"], dev))
        pk["attn_guard"] = guard
        # a WEIGHT that had to be clamped stays clamped in its planes: kept as a host-side flag of this pack (the device word is zeroed with every
        # refusal, the planes are not repacked by that: ADVICE r05) -- check_attention_guard() goes wide / refuses on it until the pack is dropped
        pk["weights_clamped"] = float(guard[depth].item())
        guard[depth:].zero_()
        pk["wide"] = self._wide()
        pk["inter_pairs"] = sorted(self._inter_pair_set())
        return pk

    def _pack_impl(self, dev):
        cfg = self.cfg
        h8_sites = self._h8_sites()
        sd = self._pack_state_dict(dev)
        pk = {"h8_sites": h8_sites, "h8c": self._h8c_wanted()}
        D = cfg["embed_dim"]

        def planes(w2d, kpad=None, fmt=ops.FMT_B3):
            return ops.split_planes(w2d.contiguous(), kpad, fmt=fmt, weight=fmt == ops.FMT_H8)

        def scalar(k):
            return float(sd[k].item())

        # --- ViT
        pk["pe_w"] = planes(sd["patch_embed.proj.weight"].reshape(D, -1))  # K order (c,kh,kw)
        pk["pe_b"] = sd["patch_embed.proj.bias"].contiguous()
        pk["blocks"] = []
        # Head widths that are not a multiple of the 32-wide MFMA k-block (ViT-H: 1280 / 16 = 80) run ZERO-PADDED per head
        # (80 -> 96): the qkv projection gets zero rows (and zero bias) for the pad channels, the output projection zero
        # columns, the rel-pos tables zero columns.  q.k, the rel-pos terms and the proj output are unchanged -- every added
        # term is an exact zero -- and the softmax scale stays head_dim^-0.5 of the true width.
        heads_ = cfg["num_heads"]
        hd_true = D // heads_
        hd_ = ops.pad32(hd_true)
        Da = heads_ * hd_            # attention width (= D unless padded)
        self._hd_true, self._hd_pad = hd_true, hd_

        def pad_head_rows(w, groups):   # [groups*heads*hd_true, ...] -> [groups*heads*hd_, ...]
            if hd_ == hd_true:
                return w.contiguous()
            v = w.reshape(groups, heads_, hd_true, *w.shape[1:])
            out = v.new_zeros(groups, heads_, hd_, *w.shape[1:])
            out[:, :, :hd_true] = v
            return out.reshape(groups * heads_ * hd_, *w.shape[1:]).contiguous()

        def pad_cols(t):                # [L, hd_true] -> [L, hd_]
            if hd_ == hd_true:
                return t
            out = t.new_zeros(t.shape[0], hd_)
            out[:, :hd_true] = t
            return out

        # operand format of the ViT-block GEMMs: h8 needs every contraction length (D, attention width, MLP hidden) % 64 == 0
        hidden_ = sd["blocks.0.mlp.lin1.weight"].shape[0]
        # ... as h8c planes (3 bytes per element, csrc/gemm_h8c.hip: a shorter operand stream and balanced matrix phases) where every contraction of
        # the block is at least 512 deep (few k-tile pairs per output tile leave that kernel's straight-line loop nothing to run), h8 line planes otherwise
        vfmt = self._pair_fmt()   # the hi/lo pair format of the ViT blocks (h8 off, or a block moved off fp16 attention): fp16 hi/lo pairs, like the attention kernels' own (bf16 pairs in the wide-range state)
        if "vit" in h8_sites and D % 64 == 0 and Da % 64 == 0 and hidden_ % 64 == 0:
            vfmt = ops.FMT_H8C if (min(D, Da, hidden_) >= 512 and self._h8c_wanted()) else ops.FMT_H8
        pk["vit_fmt"] = vfmt
        # LayerNorm fold (IE:396-421; LAB_NOTES.md 4.2): norm1 / norm2 of the ViT blocks live in their consumer GEMMs.  The producer of the
        # residual stream (proj, lin2, the injector's output projection) also writes the stream as planes and per-row strip sums; qkv / lin1
        # run on W o w and compute rstd * (x W'^T - mean * colsum(W')) + (W b + bias) in their epilogue.  tools/lnfold_study.py: same error as
        # LayerNorm + GEMM on the seeded weights, + 1.5e-5 at |mean| = 4.5 std.  Needs whole 128-column tiles and 64-column strips.
        fold = self._fold_ln_wanted(hidden_)
        pk["fold_ln"] = fold

        for i in range(cfg["depth"]):
            b = f"blocks.{i}."
            qkv_bias = pad_head_rows(sd[b + "attn.qkv.bias"], 3)
            pk["blocks"].append(dict(
                n1w=sd[b + "norm1.weight"], n1b=sd[b + "norm1.bias"], n2w=sd[b + "norm2.weight"], n2b=sd[b + "norm2.bias"],
                qkv_b=qkv_bias,
                # k = v of pad tokens; with the fp16 P V of the attention kernels (the "attnv" site, a default; `h8_sites` without it: fp16 hi/lo pairs) the v third
                # of the qkv planes -- these bias rows and the qkv GEMM's output -- is h8-encoded (ops.Planes.split)
                qkv_bp=(ops.split_planes_qkv(qkv_bias.reshape(1, -1).contiguous(), Da) if ("attnv" in h8_sites and Da % 32 == 0)
                        else ops.split_planes(qkv_bias.reshape(1, -1).contiguous(), kpad=3 * Da, fmt=ops.FMT_F3)),
                # ... and for the kernels with the rel-pos terms fused (head_dim 64) the whole row, the qkv GEMM's whole output and the
                # rel-pos tables are h8 planes: every contraction of those kernels runs on the fp16 hi parts (v_fmt = 2)
                qkv_bp16=(ops.split_planes(qkv_bias.reshape(1, -1).contiguous(), kpad=3 * Da, fmt=ops.FMT_H8) if "attnv" in h8_sites else None),
                # ... and plain bf16 hi/lo planes for a block whose logit range rules fp16 operands out (attention_precision, _attn_mode)
                qkv_bp_b3=ops.split_planes(qkv_bias.reshape(1, -1).contiguous(), kpad=3 * Da, fmt=ops.FMT_F3),
                proj_b=sd[b + "attn.proj.bias"], lin1_b=sd[b + "mlp.lin1.bias"], lin2_b=sd[b + "mlp.lin2.bias"],
                rph=pad_cols(sd[b + "attn.rel_pos_h"]), rpw=pad_cols(sd[b + "attn.rel_pos_w"]),
                ws=0 if i in cfg["global_attn_indexes"] else cfg["window_size"], index=i))
            pk["blocks"][-1].update(self._block_gemm_planes(sd, i, vfmt, fold, dev))
        for blk in pk["blocks"]:   # windowed blocks with head_dim 64: rel-pos tables packed for the fused window kernel
            wsz = blk["ws"]
            if wsz and wsz <= 14 and hd_ == 64:
                L = 2 * wsz - 1
                th = blk["rph"] if blk["rph"].shape[0] == L else _linear_resize_rows(blk["rph"], L)
                tw = blk["rpw"] if blk["rpw"].shape[0] == L else _linear_resize_rows(blk["rpw"], L)
                blk["relp"] = ops.window_relpos_planes(th, tw, wsz, fmt=ops.FMT_F3)
                blk["relp16"] = ops.window_relpos_planes(th, tw, wsz, fmt=ops.FMT_H8) if blk["qkv_bp16"] is not None else None
        # --- TwinConvNeXt
        cnx_f16 = self._cnx_f16_wanted()

        def cfmt(w2d, pw=0):   # "cnx" in h8_sites (NOT a default: the 36-block chain is the error-sensitive part of the path, LAB_NOTES.md section 2):
            # h8 operands for the pointwise convs whose two contraction lengths (C and 4C) are multiples of 64 and that do not run on the fused stage-0 kernel
            # ("cnx2": stage 2 alone -- the selective variant of tools/cnx_h8_study.py, measured and left off: LAB_NOTES.md section 4.2)
            c_ = min(w2d.shape)
            on = "cnx" in h8_sites or ("cnx2" in h8_sites and c_ == self.channels[2])
            if on and c_ % 64 == 0 and not ops.convnext_mlp_fused_supported(c_):
                return ops.FMT_H8
            # "cnx2p2" (round 6; NOT a default): pointwise_conv2 of stage 2 alone (K = 4C = 1536: 12 k-tile pairs, the h8c kernel's regime) on
            # h8c planes -- two matrix units per product instead of three; pointwise_conv1 stays on f3 and writes its GELU output as h8c planes.
            # tools/cnx_precision_trade.py + profiles/r06_cnx_precision_trade.txt: f3 / f4 error x 2 / x 3.5 at ViT-B on the oracle for the step time measured there
            if "cnx2p2" in h8_sites and pw == 2 and c_ == self.channels[2] and (4 * c_) % 64 == 0 and 4 * c_ >= 512 and self._h8c_wanted():
                return ops.FMT_H8C
            # default (round 4): fp16 hi/lo pairs ("f3": 22 significant bits, the same three MFMAs per product as bf16 hi/lo) for the whole
            # chain (stem, downsample and pointwise convs) -- its operand rounding is what GFFM amplifies (LAB_NOTES.md section 2, tools/f3_study.py)
            return ops.FMT_F3 if cnx_f16 else ops.FMT_B3
        # ConvNeXt LayerNorm fold (round 3): OPT-IN (`model.fold_convnext_ln = True`).  Built, tested, and measured at
        # ViT-L 1024^2: step -0.17 ms, golden probes 2.9e-4 -> 4.6e-4 (the chain's error is amplified ~15 x by GFFM, LAB_NOTES.md sections 2 and 4.2):
        # not worth the margin.  Applies to the stages that run pointwise_conv1 as a GEMM (not the fused stage-0 pair), 64-channel chunks.
        want_cnx = bool(getattr(self, "fold_convnext_ln", False))
        pk["cnx_f16"] = cnx_f16

        def fold_cnx(c_):
            return want_cnx and c_ % 64 == 0 and (4 * c_) % 128 == 0 and not ops.convnext_mlp_fused_supported(c_)
        pk["fold_cnx_ln"] = want_cnx
        t = "spm.twin_conv."
        pk["twin"] = {}
        for s in ("x", "y"):
            d = t + f"downsample_layers_{s}."
            st = dict(stem=planes(sd[d + "0.0.weight"].reshape(self.channels[0], -1), fmt=ops.FMT_F3 if cnx_f16 else ops.FMT_B3), stem_b=sd[d + "0.0.bias"],
                      stem_nw=sd[d + "0.1.weight"], stem_nb=sd[d + "0.1.bias"], ds=[], stages=[], out_norm=[])
            for i in range(1, 4):
                w = sd[d + f"{i}.1.weight"]  # [Cout, Cin, 2, 2] -> K order (kh, kw, cin)
                st["ds"].append(dict(nw=sd[d + f"{i}.0.weight"], nb=sd[d + f"{i}.0.bias"],
                                     w=planes(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), fmt=ops.FMT_F3 if cnx_f16 else ops.FMT_B3),
                                     b=sd[d + f"{i}.1.bias"]))
            for i in range(4):
                blks = []
                for j in range(self.depths[i]):
                    b = t + f"stages_{s}.{i}.{j}."
                    dw = sd[b + "depthwise_conv.weight"]  # [C,1,7,7] -> tap-major [49, C]
                    blk = dict(dw=dw.reshape(dw.shape[0], 49).t().contiguous(), dw_b=sd[b + "depthwise_conv.bias"],
                               nw=sd[b + "norm.weight"], nb=sd[b + "norm.bias"],
                               pw1=planes(sd[b + "pointwise_conv1.weight"], fmt=cfmt(sd[b + "pointwise_conv1.weight"])), pw1_b=sd[b + "pointwise_conv1.bias"],
                               pw2=planes(sd[b + "pointwise_conv2.weight"], fmt=cfmt(sd[b + "pointwise_conv2.weight"], 2)), pw2_b=sd[b + "pointwise_conv2.bias"],
                               gamma=sd[b + "gamma"])
                    if fold_cnx(self.channels[i]):
                        # the block's LayerNorm (TC:103-106) folded into pointwise_conv1: W o w as planes, their column sums as the kernel
                        # reads them, W b + bias (the depthwise conv writes the RAW planes + strip sums, _spm)
                        w1 = sd[b + "pointwise_conv1.weight"]
                        pl = planes(w1 * sd[b + "norm.weight"][None, :], fmt=cfmt(w1))
                        blk["pw1f"] = pl
                        blk["pw1_cs"] = ops.planes_to_float(pl, cols=w1.shape[1])[: w1.shape[0]].double().sum(1).float().contiguous()
                        blk["pw1_bf"] = (w1.double() @ sd[b + "norm.bias"].double()).float().add_(sd[b + "pointwise_conv1.bias"]).contiguous()
                    blks.append(blk)
                st["stages"].append(blks)
                st["out_norm"].append((sd[t + f"norm_{s}{i}.weight"], sd[t + f"norm_{s}{i}.bias"]))
            pk["twin"][s] = st
        # both streams stacked along a leading batch dimension: the two ConvNeXt chains run as ONE chain of batched kernels
        # (per-batch weights), see _twin_batched
        def stack2(fx, fy):
            if isinstance(fx, ops.Planes):
                buf = torch.cat([fx.p, fy.p], 0).contiguous()
                pl = ops.Planes(buf[:fx.p.shape[0]], fx.n, fx.k, fx.kpad, fx.fmt, fx.weight)    # (h8c planes hold row PAIRS: n / 2 tensor rows)
                pl.full = buf           # keeps the second batch alive; batch stride = Planes.batch_stride(n)
                return pl
            return torch.stack([fx, fy], 0).contiguous()

        def merge(ax, ay):
            if isinstance(ax, dict):
                return {k: merge(ax[k], ay[k]) for k in ax}
            if isinstance(ax, (list, tuple)):
                return [merge(u, v) for u, v in zip(ax, ay)]
            return stack2(ax, ay)
        pk["twin2"] = merge(pk["twin"]["x"], pk["twin"]["y"])
        del pk["twin"]     # only the stacked form is used by the forward (and written by checkpoint.save_packed)
        # --- fusion neck
        f = "spm.smart_fusion."
        pk["neck"] = []
        for i in range(4):
            C = 2 * self.channels[i]
            c = self.channels[i]
            lv = dict(gfe=[], loc=[])
            for m in ("rgb", "sne"):
                b = f + f"global_feature_encoder_{m}.{i}."
                q1 = sd[b + "attn.qkv1.weight"]  # [3c, c/32, 1, 1], groups 32 -> [G][tap][ci][co]
                q2 = sd[b + "attn.qkv2.weight"]  # [3c, 3c/32, 3, 3]
                G = 32
                co1, ci1 = 3 * c // G, c // G
                co2, ci2 = 3 * c // G, 3 * c // G
                lv["gfe"].append(dict(
                    nw=sd[b + "norm1.body.weight"], nb=sd[b + "norm1.body.bias"],
                    q1=q1.reshape(G, co1, ci1, 1).permute(0, 3, 2, 1).contiguous(),
                    q2=q2.reshape(G, co2, ci2, 9).permute(0, 3, 2, 1).contiguous(),
                    temp=sd[b + "attn.scale"].reshape(8).contiguous(), scale2=scalar(b + "attn.scale2"),
                    proj=sd[b + "attn.proj.weight"].reshape(c, c).contiguous()))
                b = f + f"local_feature_encoder_{m}.{i}."
                dw = sd[b + "bottleneckBlock.2.weight"]
                lv["loc"].append(dict(
                    w1=planes(sd[b + "bottleneckBlock.0.weight"].reshape(2 * c, c)),
                    dw=dw.reshape(2 * c, 9).t().contiguous(),
                    w3=planes(sd[b + "bottleneckBlock.4.weight"].reshape(c, 2 * c)), scale=scalar(b + "scale")))
            b = f + f"fuse_blocks.{i}."
            lv["gx"], lv["gy"] = scalar(b + "gammax.scale"), scalar(b + "gammay.scale")
            lv["lnw"], lv["lnb"] = sd[b + "norm.weight"].contiguous(), sd[b + "norm.bias"].contiguous()
            lv["lnw_mean"], lv["lnb_mean"] = float(lv["lnw"].double().mean().item()), float(lv["lnb"].double().mean().item())
            b = f + f"enhance_blocks.{i}.conv_atten."
            lv["ffrm_w"] = sd[b + "conv.weight"].reshape(C, C).contiguous()
            lv["gn_w"], lv["gn_b"] = sd[b + "gn.weight"], sd[b + "gn.bias"]
            b = f + f"detail_feature_extractions.{i}."
            dw = sd[b + "dwconv.weight"]  # [2C, 2, 3, 3], groups=C -> [G=C][tap][ci=2][co=2]
            lv["mlp_in"] = planes(sd[b + "project_in.weight"].reshape(2 * C, C))
            lv["mlp_dw"] = dw.reshape(C, 2, 2, 9).permute(3, 0, 2, 1).contiguous()   # tap-major [9][g][ci][co]
            lv["mlp_out"] = planes(sd[b + "project_out.weight"].reshape(C, C))
            lv["s1"], lv["s2"] = scalar(f + f"scale_layers.{i}.scale1"), scalar(f + f"scale_layers.{i}.scale2")
            b = f + f"ca_blocks.{i}.coord_atten."
            mip = max(8, C // 32)
            inv = sd[b + "bn1.weight"] / torch.sqrt(sd[b + "bn1.running_var"] + 1e-5)  # BN(eval) folded into conv1
            w1 = sd[b + "conv1.weight"].reshape(mip, C) * inv[:, None]
            b1 = (sd[b + "conv1.bias"] - sd[b + "bn1.running_mean"]) * inv + sd[b + "bn1.bias"]
            lv["mip"], lv["mip_pad"] = mip, ops.pad32(mip)
            lv["ca1"], lv["ca1_b"] = planes(w1), b1.contiguous()
            lv["cah"], lv["cah_b"] = planes(sd[b + "conv_h.weight"].reshape(C, mip)), sd[b + "conv_h.bias"]
            lv["caw"], lv["caw_b"] = planes(sd[b + "conv_w.weight"].reshape(C, mip)), sd[b + "conv_w.bias"]
            lv["fc"] = planes(sd[f"spm.fc{i+1}.weight"].reshape(D, C))
            fb = sd[f"spm.fc{i+1}.bias"]
            lv["fc_b"] = (fb + sd["level_embed"][i - 1]).contiguous() if i >= 1 else fb.contiguous()  # BK:149-156 folded
            pk["neck"].append(lv)
        # --- interactions
        M, Pn = cfg["deform_num_heads"], cfg["n_points"]

        inter_pairs = self._inter_pair_set()
        cur_inter = [None]      # index of the interaction being packed (pack_msda / pack_extractor are called inside the loop below)

        def ifmt(w2d, site="inter", h8c_ok=True):   # operand format of one GEMM of a site group: h8c / h8 when selected and the contraction length allows it
            kk = ops.pad32(w2d.shape[1])
            if site == "inter" and cur_inter[0] in inter_pairs and site in h8_sites:
                return self._pair_fmt()             # this interaction follows its blocks onto hi/lo pairs (inter_follow_blocks)
            if site not in h8_sites or kk % 64:
                return ops.FMT_B3
            return ops.FMT_H8C if (h8c_ok and kk >= 512 and self._h8c_wanted()) else ops.FMT_H8

        def iplanes(w2d, site="inter", h8c_ok=True):
            return planes(w2d, fmt=ifmt(w2d, site, h8c_ok))

        # Injector i normalises the adapter tokens c with its feat_norm, extractor i normalises the SAME c (the injector only
        # updates x) with its query_norm: two LayerNorm passes over the largest token matrix of the path (Nc = 21504 rows per
        # image).  LN(c; w, b) W^T + bias = chat (W * w)^T + (W b + bias) with chat = (c - mean) * rstd, so the affine parts are folded
        # into the two projections' weights here and ONE pass computes chat for both (`share_c_norm = False` keeps two).
        share_c = bool(getattr(self, "share_c_norm", True))
        pk["share_c_norm"] = share_c
        pk["ln_one"], pk["ln_zero"] = torch.ones(D, device=dev), torch.zeros(D, device=dev)

        def fold_ln(w2d, bias, lnw, lnb):      # -> weight and bias of the projection applied to the un-affine normalised tokens
            return (w2d * lnw[None, :]).contiguous(), (bias + w2d @ lnb).contiguous()

        # Round 5: the adapter tokens' LayerNorms inside their consumer GEMMs (`fold_adapter_ln`; the ViT blocks' fold of round 3 applied to AM:490-511,525-542).
        # Every LayerNorm over c -- the shared injector feat_norm / extractor query_norm, the extra extractors' query_norm, every ffn_norm -- follows a GEMM that
        # has just written c (ConvFFN fc2, the extractor's output projection).  That producer also writes c's RAW operand planes and per-row strip sums;
        # mmsa_rowstats_finalize turns them into (mean, rstd); the consumers (value / offsets projections, fc1) run on W o w with the row-normalising epilogue.
        # Twelve passes over the largest token matrix of the path (21504 x 1024 per image: 88 MB read + 66 MB planes written each) go away.  Needs the affine
        # parts folded into the weights (share_c_norm) and whole 128-column tiles at every consumer; tools/lnfold_adapter_study.py: no measurable error.
        def colsum(pl):   # column sums of packed planes as the kernel reads them (what every x_k is actually multiplied with)
            return ops.planes_to_float(pl, cols=pl.k)[: pl.n].double().sum(1).float().contiguous()
        hid_c = int(D * cfg["cffn_ratio"]) if cfg["with_cffn"] else 128
        # OPT-IN (`model.fold_adapter_ln = True`): built, pinned against the oracle (tests/test_backbone_gpu.py::test_adapter_layernorm_fold_against_the_oracle), and
        # measured step-NEUTRAL at ViT-L 1024^2 (profiles/r05_adapter_ln_fold.txt: 31.12 / 31.39 ms without, 31.41 / 31.34 ms with; golden probes 0.94e-4 ->
        # 1.15e-4): the eleven passes it deletes cost what the producers' extra plane + strip-sum stores and the consumers' row loads cost -- by bytes the fold
        # saves only the read of c (176 MB per pass at batch 2), 0.3 ms at best.
        fold_rn = self._fold_adapter_ln_wanted()
        pk["fold_adapter_ln"] = fold_rn

        def pack_msda(b, fold_val=None, fold_oa=None):
            w = torch.cat([sd[b + "sampling_offsets.weight"], sd[b + "attention_weights.weight"]], 0)
            bb = torch.cat([sd[b + "sampling_offsets.bias"], sd[b + "attention_weights.bias"]], 0)
            wv, bv_ = sd[b + "value_proj.weight"], sd[b + "value_proj.bias"]
            if fold_oa is not None:
                w, bb = fold_ln(w, bb, *fold_oa)
            if fold_val is not None:
                wv, bv_ = fold_ln(wv, bv_, *fold_val)
            # the offsets + attention-weights projection has 3 * heads * levels * points output columns (192 / 576 at ViT-L): padded with zero rows to whole
            # 128-column GEMM tiles -- the same number of tiles, none of them ragged, so that every tile of the launch takes the register-resident epilogue
            # (a launch that runs both epilogues no longer fits the instruction cache: profiles/r05_epilogue_regs.txt).  The deformable-attention kernel
            # reads the first 3 * heads * levels * points columns of a row (ldraw = the padded width).
            n_oa = w.shape[0]
            if n_oa % 128:
                npad = 128 - n_oa % 128
                w = torch.cat([w, w.new_zeros(npad, w.shape[1])], 0)
                bb = torch.cat([bb, bb.new_zeros(npad)], 0)
            mp = dict(oa=iplanes(w), oa_b=bb.contiguous(), val=iplanes(wv), val_b=bv_,
                      out=iplanes(sd[b + "output_proj.weight"]), out_b=sd[b + "output_proj.bias"])
            if fold_rn:
                if fold_oa is not None:
                    mp["oa_cs"] = colsum(mp["oa"])
                if fold_val is not None:
                    mp["val_cs"] = colsum(mp["val"])
            return mp

        def pack_extractor(b, fold_c=False):
            ep = dict(qnw=sd[b + "query_norm.weight"], qnb=sd[b + "query_norm.bias"], fnw=sd[b + "feat_norm.weight"],
                      fnb=sd[b + "feat_norm.bias"], fold_c=fold_c, first=False, cffn=self.cfg["with_cffn"],
                      attn=pack_msda(b + "attn.", fold_oa=(sd[b + "query_norm.weight"], sd[b + "query_norm.bias"]) if fold_c else None))
            if self.cfg["with_cffn"]:   # AM:485-488, 499-500
                dw = sd[b + "ffn.dwconv.dwconv.weight"]
                fc1_w, fc1_bias = sd[b + "ffn.fc1.weight"], sd[b + "ffn.fc1.bias"]
                if fold_c:   # ffn_norm folded into fc1 (its input is then the un-affine normalised c, or raw c + the row-normalising epilogue)
                    fc1_w, fc1_bias = fold_ln(fc1_w, fc1_bias, sd[b + "ffn_norm.weight"], sd[b + "ffn_norm.bias"])
                ep.update(fc1=iplanes(fc1_w), fc1_b=fc1_bias,
                          dw=dw.reshape(dw.shape[0], 9).t().contiguous(), dw_b=sd[b + "ffn.dwconv.dwconv.bias"],
                          # fc2's A operand is written by the 3 x 3 depthwise conv (mmsa_dwconv_nhwc), which emits bf16 hi/lo and h8 LINE planes only:
                          # a hidden width >= 512 (cffn_ratio 0.5 at embed_dim 1024; the reference's configs stay <= 320) must not pick h8c here
                          fc2=iplanes(sd[b + "ffn.fc2.weight"], h8c_ok=False), fc2_b=sd[b + "ffn.fc2.bias"],
                          ffw=sd[b + "ffn_norm.weight"], ffb=sd[b + "ffn_norm.bias"])
                if fold_rn and fold_c:
                    ep["fc1_cs"] = colsum(ep["fc1"])
            return ep

        pk["inter"] = []
        n_int = len(self.interaction_indexes)
        for i in range(n_int):
            b = f"interactions.{i}."
            cur_inter[0] = i
            it = dict(inj=dict(gamma=sd[b + "injector.gamma"], qnw=sd[b + "injector.query_norm.weight"],
                               qnb=sd[b + "injector.query_norm.bias"], fnw=sd[b + "injector.feat_norm.weight"],
                               fnb=sd[b + "injector.feat_norm.bias"], fold_c=share_c,
                               attn=pack_msda(b + "injector.attn.", fold_val=(sd[b + "injector.feat_norm.weight"], sd[b + "injector.feat_norm.bias"]) if share_c else None)),
                      ext=[pack_extractor(b + "extractor.", fold_c=share_c)])
            it["ext"][0]["first"] = True     # the extractor that shares the injector's un-affine normalised c (share_c_norm) when nothing is folded further
            if i == n_int - 1 and self.cfg["use_extra_extractor"]:   # BK:91-92
                it["ext"] += [pack_extractor(b + "extra_extractors.0.", fold_c=share_c), pack_extractor(b + "extra_extractors.1.", fold_c=share_c)]
            it["inj"]["first_block"] = self.interaction_indexes[i][0]   # whose qkv GEMM reads the stream planes the injector writes (LayerNorm fold)
            pk["inter"].append(it)
        cur_inter[0] = None
        # --- tail: ConvTranspose2d(D,D,2,2) weight [Cin, Cout, 2, 2] -> rows (i,j,co), K = ci  (BK:55,324)
        up = sd["up.weight"]
        pk["up"] = iplanes(up.permute(2, 3, 1, 0).reshape(4 * D, D), "up")
        pk["up_b"] = sd["up.bias"].repeat(4).contiguous()
        pk["bn"] = []
        for i in range(1, 5):
            inv = sd[f"norm{i}.weight"] / torch.sqrt(sd[f"norm{i}.running_var"] + 1e-5)
            pk["bn"].append((inv.contiguous(), (sd[f"norm{i}.bias"] - sd[f"norm{i}.running_mean"] * inv).contiguous()))
        pk["pos_src"] = sd["pos_embed"]
        pk["geom"] = {}
        torch.cuda.synchronize(dev)
        return pk

    # geometry-dependent, input-independent tables (cached per (H, W))
    @torch.no_grad()
    def _geometry(self, H, W, dev):
        pk = self._packed
        key = (H, W)
        if key in pk["geom"]:
            return pk["geom"][key]
        cfg = self.cfg
        p = cfg["patch_size"]
        Hp, Wp = H // p, W // p
        g = {}
        g["pos"] = _bicubic_resize(pk["pos_src"][0], Hp, Wp).reshape(Hp * Wp, -1).contiguous()  # BK:136-143
        # deform_inputs (AM:397-431)
        ss1 = torch.tensor([(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)], dtype=torch.int64)
        ss2 = torch.tensor([(H // 16, W // 16)], dtype=torch.int64)
        g["ss1"], g["ss2"] = ss1.to(dev), ss2.to(dev)
        g["lsi1"] = torch.cat((ss1.new_zeros(1), ss1.prod(1).cumsum(0)[:-1])).to(dev)
        g["lsi2"] = ss2.new_zeros(1).to(dev)
        g["ref1"] = _ref_points([(H // 16, W // 16)]).to(dev)
        g["ref2"] = _ref_points([(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]).to(dev)
        # rel-pos gather tables (IE:554-584)
        g["rel"] = []
        ws = cfg["window_size"]
        hd_ = self._hd_pad
        g["relg"] = []
        for blk in pk["blocks"]:
            relg = None
            if blk["ws"]:
                g["rel"].append((_rel_table(ws, blk["rph"]), _rel_table(ws, blk["rpw"])) if blk.get("relp") is None else None)
            elif hd_ == 64 and Wp == 64 and Hp <= 64 and Hp % 4 == 0:
                # global block on a 64-wide grid: rel-pos terms computed inside the attention kernel from the packed tables
                th = blk["rph"] if blk["rph"].shape[0] == 2 * Hp - 1 else _linear_resize_rows(blk["rph"], 2 * Hp - 1)
                tw = blk["rpw"] if blk["rpw"].shape[0] == 2 * Wp - 1 else _linear_resize_rows(blk["rpw"], 2 * Wp - 1)
                relg = (ops.global_relpos_planes(th, tw, fmt=ops.FMT_F3),
                        ops.global_relpos_planes(th, tw, fmt=ops.FMT_H8) if blk["qkv_bp16"] is not None else None)
                g["rel"].append(None)
            else:
                g["rel"].append((_rel_table(Hp, blk["rph"]), _rel_table(Wp, blk["rpw"])))
            g["relg"].append(relg)
        pk["geom"][key] = g
        return g

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def _prepare(self, x):
        if not x.is_cuda:
            raise RuntimeError("mmsa: input must be a GPU tensor; the MI355X backbone has no CPU path")
        if x.dim() != 4 or x.shape[1] != 6:
            raise RuntimeError(f"mmsa: expected [B, 6, H, W] (rgb + auxiliary modality), got {tuple(x.shape)}")
        B, _, H, W = x.shape
        if H != self.img_size or W != self.img_size:
            raise RuntimeError(f"mmsa: H=W=img_size={self.img_size} is required (GFFM LayerNorm length, AM:240-241); got {H}x{W}")
        if H % 32 != 0:
            raise RuntimeError("mmsa: img_size must be a multiple of 32")
        dev = x.device
        x = x.contiguous().float()
        if (self._packed is None or self._packed.get("dev") != dev or tuple(self._packed.get("h8_sites", ())) != self._h8_sites()
                or self._packed.get("h8c") != self._h8c_wanted() or self._packed.get("cnx_f16") != self._cnx_f16_wanted()
                or bool(self._packed.get("wide", False)) != self._wide() or list(self._packed.get("inter_pairs", [])) != sorted(self._inter_pair_set())):
            self._packed = self._pack(dev)
            self._packed["dev"] = dev
        if self._ws is None or self._ws.device != dev:
            self._ws = Workspace(dev)
        return x, B, H, W

    def _cbufs(self, B, H, W, tag=""):
        """c1 [B*HW/16, D] and c = (c2|c3|c4 + level embed) [B, Nc, D]; `tag` selects one of the two sets of the pipelined mode."""
        D = self.cfg["embed_dim"]
        Nc = (H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32)
        return self._ws.get("c1" + tag, B * (H // 4) * (W // 4), D), self._ws.get("c" + tag, B * Nc, D), Nc

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("mmsa: input must be a GPU tensor; the MI355X backbone has no CPU path")
        with torch.cuda.device(x.device):    # launches go to the current stream of the input's device
            guard = self.attention_guard == "sync" and not torch.cuda.is_current_stream_capturing()
            depth = self.cfg["depth"]
            for _ in range(depth + 2):                # (+ 1: the blocks one by one; + 1: the switch to the wide-range state)
                x, B, H, W = self._prepare(x)         # (packs on the first pass -- and again after the range fallback dropped the pack)
                with ops.clamp_watch(self._packed["attn_guard"][depth:]):   # every plane-producing launch reports values beyond its format's range
                    # ---- spatial prior module -> c1, c
                    c1, cbuf, Nc = self._cbufs(B, H, W)
                    c1_ready = self._spm(x, B, H, W, c1, cbuf, Nc)
                    outs = self._vit(x, B, H, W, c1, cbuf, c1_ready)
                # ---- attention logit guard: a block that ran fp16 attention beyond its range has been moved to fp16 hi/lo pairs -> once more;
                # clamp watch: a value beyond an fp16-based format's range -> the model is in its wide-range state now (repacked by _prepare) -> once more
                if not guard or not self.check_attention_guard():
                    break
            return outs, None

    @torch.no_grad()
    def forward_taps(self, x):
        """forward() that also returns copies of the intermediate tensors the oracle taps (oracle/ref_encoder.py OracleEncoder.forward
        `taps`): twin{i} / fuse{i} (NCHW), c1, c_in, x_in, x{i}, c{i} ([B, N, D]).  Test aid of the per-stage error budget
        (tests/test_backbone_gpu.py::test_vitl1024_error_budget); the copies are plain device-to-device clones on the launch streams."""
        self._taps = {}
        try:
            outs, _ = self.forward(x)
            torch.cuda.synchronize(x.device)
            taps = self._taps
        finally:
            self._taps = None
        B, _, H, W = x.shape
        D = self.cfg["embed_dim"]
        res = {}
        for k, v in taps.items():
            if k.startswith(("twin", "fuse")):
                i = int(k[-1])
                h, w = H // (4 << i), W // (4 << i)
                res[k] = v.reshape(B, h, w, -1).permute(0, 3, 1, 2).contiguous()
            else:
                res[k] = v.reshape(B, -1, D)
        for i, f in enumerate(outs):
            res[f"f{i + 1}"] = f
        return outs, res

    def forward_pipelined(self, x_next):
        """Throughput mode (two-stage software pipeline over consecutive batches).  The spatial prior module depends only on
        the image, and every ViT block depends on it (the first injector reads c2..c4), so inside one batch the two cannot
        overlap -- but the SPM of the NEXT batch can run underneath the ViT blocks of the current one: its many small launches
        fill the CUs that the ViT's kernels leave idle.  Each call starts the SPM of `x_next` on a side stream (into the other
        of two c1 / c buffer sets) and runs the ViT + interactions + tail of the batch given to the PREVIOUS call, whose
        outputs it returns (None on the first call; pass None to drain).  Per batch the arithmetic and the results are those of
        forward(), bit for bit."""
        cur = getattr(self, "_pl_cur", None)
        dev_ = x_next.device if x_next is not None else (cur["x"].device if cur is not None else None)
        if dev_ is None:
            return None
        with torch.cuda.device(dev_):
            return self._forward_pipelined(x_next, cur)

    def _forward_pipelined(self, x_next, cur):
        main = torch.cuda.current_stream()
        nxt = None
        if x_next is not None:
            x_next, B, H, W = self._prepare(x_next)
            q = 1 - cur["set"] if cur is not None else 0
            c1, cbuf, Nc = self._cbufs(B, H, W, f"_p{q}")
            if getattr(self, "_pipe_stream", None) is None or self._pipe_stream.device != x_next.device:
                self._pipe_stream = torch.cuda.Stream(device=x_next.device)
            sa = self._pipe_stream
            ops.zero_(self._ws.get("pl_tick", 1, 1))   # a node before the fork: forking off a capture stream that has recorded nothing yet crashed hipStreamEndCapture
            sa.wait_stream(main)
            with torch.cuda.stream(sa):
                joins = self._spm(x_next, B, H, W, c1, cbuf, Nc, join=False)
            nxt = dict(x=x_next, set=q, dims=(B, H, W), c1=c1, c=cbuf)
        outs = None
        if cur is not None:
            B, H, W = cur["dims"]
            outs = self._vit(cur["x"], B, H, W, cur["c1"], cur["c"], None)
        if nxt is not None:
            # join: the next call (or graph replay) starts after both stages.  Every neck stream is joined by THIS stream, not by
            # the SPM's: under HIP-graph capture a forked stream that waits on another forked stream's event crashed
            # hipStreamEndCapture (ROCm 7.2; tools/cap_min3.py), the capture's origin stream may.
            for e in joins:
                main.wait_event(e)
            main.wait_stream(self._pipe_stream)
        self._pl_cur = nxt
        return (outs, None) if outs is not None else None

    def _vit(self, x, B, H, W, c1, cbuf, c1_ready):
        pk, ws, cfg = self._packed, self._ws, self.cfg
        dev = x.device
        geo = self._geometry(H, W, dev)
        D = cfg["embed_dim"]
        p = cfg["patch_size"]
        Hp, Wp = H // p, W // p
        T = Hp * Wp
        n2, n3, n4 = (H // 8) * (W // 8), T, (H // 32) * (W // 32)
        Nc = n2 + n3 + n4

        # ---- patch embedding + absolute position embedding (IE:662-671, BK:268-278)
        a = ws.get("pe_a", B * T, pk["pe_w"].kpad)
        ops.im2col_nchw(x, 0, 3, p, a)
        xs = [ws.get(f"x{i}", B * T, D) for i in range(len(self.interaction_indexes) + 1)]
        ops.gemm(a, pk["pe_w"], xs[0], bias=pk["pe_b"], resid=geo["pos"], resid_mod=T)
        taps = self._taps
        if taps is not None:
            taps["x_in"], taps["c_in"] = xs[0].clone(), cbuf[:B * Nc].clone()

        # ---- interactions (AM:567-581)
        self._cfold = None     # adapter-token LayerNorm fold: no producer GEMM has written c yet (the first interaction normalises c with a LayerNorm pass)
        n_inter = len(self.interaction_indexes)
        for i, idx in enumerate(self.interaction_indexes):
            it = pk["inter"][i]
            self._injector(it["inj"], xs[i], xs[i + 1], cbuf, geo, B, T, Nc)   # (with the LayerNorm fold: also the producer of block idx[0]'s stream planes)
            for bi in range(idx[0], idx[-1] + 1):
                self._block(pk["blocks"][bi], geo["rel"][bi], xs[i + 1], B, Hp, Wp, geo["relg"][bi],
                            next_fmt=pk["blocks"][bi + 1]["qkv"].fmt if bi < idx[-1] else None)
            for j, ex in enumerate(it["ext"]):
                self._extractor(ex, cbuf, xs[i + 1], geo, B, T, Nc, H, W, last=(i == n_inter - 1 and j == len(it["ext"]) - 1))
            if taps is not None:
                taps[f"x{i}"], taps[f"c{i}"] = xs[i + 1].clone(), cbuf[:B * Nc].clone()
        self._cfold = None

        # ---- tail (BK:316-337)
        if c1_ready is not None:
            torch.cuda.current_stream().wait_event(c1_ready)
        if taps is not None:
            taps["c1_map"] = c1.clone()
        outs = []
        c2p = ws.planes("up_a", B * n2, D, fmt=pk["up"].fmt)
        for bi in range(B):
            ops.split_planes(cbuf[bi * Nc:bi * Nc + n2], kpad=D, out=c2p.rows(bi * n2, (bi + 1) * n2))
        ops.gemm(c2p, pk["up"], c1, bias=pk["up_b"], resid=c1, batch=B, m=n2, stride_a=c2p.batch_stride(n2),
                 stride_r=(H // 4) * (W // 4) * D, stride_c=(H // 4) * (W // 4) * D, pixel_shuffle=(H // 8, W // 8, D))
        # emit_planes (set by the decode head's caller, e.g. bench.py / mmsa.inference): every output map is also written
        # token-major as interleaved planes and attached to the returned tensor (`_mmsa_planes`), so that mmsa.SegformerHead
        # feeds its first 1x1 convs without the NCHW -> planes transposition of 0.7 GB per step
        emit = bool(getattr(self, "emit_planes", False)) and D % 32 == 0
        ctag = getattr(self._ws, "tag", "")   # "" or the chain's workspace tag (mmsa.chains): every chain has its own output planes
        if not hasattr(self, "_planes_gens"):
            self._planes_gens = {}
        pgen = self._planes_gens.setdefault(ctag, [0])
        pgen[0] += 1     # the f*_out planes of earlier calls (of this chain) are overwritten below: their holders see live() == False
        outs = []
        for k, (src, cs, (hh, wwd)) in enumerate(((c1, (H // 4) * (W // 4) * D, (H // 4, W // 4)), (cbuf, Nc * D, (H // 8, W // 8)),
                                                  (cbuf[n2:], Nc * D, (Hp, Wp)), (cbuf[n2 + n3:], Nc * D, (H // 32, W // 32)))):
            f = torch.empty(B, D, hh, wwd, device=dev)
            fpl = ws.planes(f"f{k + 1}_out", B * hh * wwd, D) if emit else None
            ops.tail_fuse(src, cs, xs[k + 1] if self.cfg["add_vit_feature"] else None, *pk["bn"][k], f, B, hh, wwd, Hp, Wp, out_planes=fpl)   # BK:326-331
            if emit:
                f._mmsa_planes = fpl.stamp(pgen)
            outs.append(f)
        f1, f2, f3, f4 = outs
        return [f1, f2, f3, f4]

    # ------------------------------------------------------------------ SAM ViT block (IE:382-423)
    def _stream_planes(self, rows, fmt):
        """(planes of the residual stream in its next consumer's operand format, its per-row strip sums) -- written by the stream's producer
        GEMMs when the LayerNorms are folded"""
        D = self.cfg["embed_dim"]
        return (self._ws.planes("blk_xp", rows, D, fmt=fmt), self._ws.get("blk_rs", rows, 2 * (D // 64)))

    def _block(self, bp, rel, x, B, Hp, Wp, relg=None, next_fmt=None):
        """One SAM ViT block in place on x.  `next_fmt` (LayerNorm fold): operand format of the NEXT block's qkv GEMM, whose stream planes
        this block's lin2 then writes; None = nobody reads them (last block before an extractor)."""
        ws, cfg = self._ws, self.cfg
        D, heads = cfg["embed_dim"], cfg["num_heads"]
        hd = self._hd_pad                 # head width the kernels see (zero-padded to a multiple of 32: _pack)
        scale = self._hd_true ** -0.5     # IE:445: head_dim ** -0.5 of the model's true width
        Da = heads * hd
        T = Hp * Wp
        # LayerNorm fold: norm1 / norm2 inside the qkv / lin1 GEMMs; x's planes and row sums come from its producer.  The row-normalising
        # epilogue lives in the 256-row LDS-DMA kernel only (M >= 128): a forward with fewer token rows (one 128 x 128 image, a small slide
        # crop) runs the SAME folded weights behind a plain LayerNorm pass without affine part -- LN(x; w, b) W^T + bias = xhat (W o w)^T + (W b + bias)
        folded_w = self._packed["fold_ln"]
        fold = folded_w and B * T >= 128
        # intermediate activations travel as bf16 hi/lo planes: split once by the producer, consumed by the
        # GEMM / attention kernels with plain 16-byte copies (same bytes as fp32, no re-splitting per column block)
        fused = bp.get("relp") is not None or relg is not None
        # fp16 operands inside the attention kernels ("attnv" site) only where this block's logits are small enough: _attn_mode (on the
        # first forward it may also move this block's qkv GEMM to bf16 hi/lo operands)
        f16 = bp["qkv_bp16"] is not None and self._attn_mode(bp) == "f16"
        vf = bp["qkv"].fmt   # operand format of this block's qkv GEMM: its producer wrote the stream planes / LayerNorm writes them in it
        gw = self._packed["attn_guard"][bp["index"]:bp["index"] + 1]   # this block's logit guard word
        if fold:
            xp, rs = self._stream_planes(B * T, vf)
            mr = ws.get("blk_mr", B * T, 2)
            ops.rowstats_finalize(rs, B * T, D, 1e-6, mr)
            n = None
        else:
            n = ws.planes("blk_n", B * T, D, fmt=vf)
            if folded_w:
                ops.layernorm(x, self._packed["ln_one"], self._packed["ln_zero"], 1e-6, out_planes=n)
            else:
                ops.layernorm(x, bp["n1w"], bp["n1b"], 1e-6, out_planes=n)
        all16 = fused and f16            # the fused kernels: every contraction on fp16 hi parts of h8 planes (v_fmt = 2), or none (0)
        qkv = ws.planes("blk_qkv", B * T, 3 * Da, fmt=ops.FMT_H8 if all16 else ops.FMT_F3)
        if all16:
            bias_p = bp["qkv_bp16"]
        elif f16 and not fused:          # the kernel with a rel-pos prepass: the qkv GEMM writes the v columns as h8 planes (fp16 P V only, v_fmt = 1)
            bias_p = bp["qkv_bp"]
            qkv.split = bias_p.split
        else:
            bias_p = bp["qkv_bp_b3"]
        if fold:
            ops.gemm(xp, bp["qkv"], bias=bp["qkv_bf"], out_planes=qkv, row_norm=(mr, bp["qkv_cs"]))
        else:
            ops.gemm(n, bp["qkv"], bias=bp["qkv_bf"] if folded_w else bp["qkv_b"], out_planes=qkv)
        wsz = bp["ws"]
        ao = ws.planes("blk_ao", B * T, Da, fmt=bp["proj"].fmt)
        if bp.get("relp") is not None:   # windowed block, head_dim 64: K/V-resident kernel with the rel-pos terms fused
            ops.window_attention(qkv, bias_p, bp["relp16"] if all16 else bp["relp"], ao, B, Hp, Wp, heads, hd, wsz, scale, max_logit=gw)
        elif relg is not None:           # global block on a 64-wide grid: flash kernel with the rel-pos terms fused
            ops.global_attention(qkv, bias_p, relg[1] if all16 else relg[0], ao, B, Hp, Wp, heads, hd, scale, max_logit=gw)
        else:
            kk = 2 * wsz if wsz else Hp + Wp
            rp = ws.get("blk_rp", B * heads * T, kk)
            ops.relpos_bias(qkv, rel[0], rel[1], rp, B, Hp, Wp, heads, hd, wsz)
            ops.attention(qkv, bias_p, rp, ao, B, Hp, Wp, heads, hd, wsz, scale, max_logit=gw)
        vf = bp["lin1"].fmt              # the MLP's operand format (the qkv GEMM of a block with large logits runs on bf16 hi/lo: _attn_mode)
        h = ws.planes("blk_h", B * T, bp["lin1"].n, fmt=vf)
        if fold:
            xp, rs = self._stream_planes(B * T, vf)
            ops.gemm(ao, bp["proj"], x, bias=bp["proj_b"], resid=x, out_planes=xp, rowstats_out=rs)
            ops.rowstats_finalize(rs, B * T, D, 1e-6, mr)
            ops.gemm(xp, bp["lin1"], bias=bp["lin1_bf"], act="gelu", out_planes=h, row_norm=(mr, bp["lin1_cs"]))
            if next_fmt is not None:   # the next block's norm1 is folded too: this GEMM is its producer
                xp, rs = self._stream_planes(B * T, next_fmt)
                ops.gemm(h, bp["lin2"], x, bias=bp["lin2_b"], resid=x, out_planes=xp, rowstats_out=rs)
            else:
                ops.gemm(h, bp["lin2"], x, bias=bp["lin2_b"], resid=x)
            return
        ops.gemm(ao, bp["proj"], x, bias=bp["proj_b"], resid=x)
        n = ws.planes("blk_n", B * T, D, fmt=vf)
        if folded_w:
            ops.layernorm(x, self._packed["ln_one"], self._packed["ln_zero"], 1e-6, out_planes=n)
        else:
            ops.layernorm(x, bp["n2w"], bp["n2b"], 1e-6, out_planes=n)
        ops.gemm(n, bp["lin1"], bias=bp["lin1_bf"] if folded_w else bp["lin1_b"], act="gelu", out_planes=h)
        ops.gemm(h, bp["lin2"], x, bias=bp["lin2_b"], resid=x)

    # fp16 operands inside the attention kernels are safe only while the logits stay small: the rounding error of q.k grows with the
    # logit's magnitude and is then exponentiated.  tools/attention_precision_study.py --logit-scale (CPU oracle, ViT-B, f1..f4 against
    # fp32): max |logit| 3.8 (the seeded test weights) -> 3e-5 for the all-fp16 kernels; 13 -> 1.5e-4; 48 -> 8e-3 (P V alone in fp16:
    # 3e-3), while bf16 hi/lo operands stay at 3e-6 / 1e-5.  Released SAM checkpoints are peakier than the seeded weights, so the
    # format is a MEASURED per-block decision, and it is measured on EVERY batch: the attention kernels fold the largest |logit| they
    # score (rel-pos terms included) into one device word per block (include/mmsa.h "Attention logit guard"; `_packed["attn_guard"]`).
    ATTN_F16_MAX_LOGIT = 8.0

    # What happens with the guard words (`model.attention_guard`):
    #   "sync" (default)  every EAGER forward reads them back (one 4*depth-byte copy, a host sync) before it returns; a block that ran fp16
    #                     attention on logits above ATTN_F16_MAX_LOGIT moves to bf16 hi/lo operands -- its attention kernels AND its four
    #                     GEMM weights, repacked from the state dict -- and the forward is run again, so the tensors returned were never
    #                     computed on fp16 attention beyond the threshold.  A block only ever moves from fp16 to bf16 hi/lo.
    #   "off"             nothing is read back; `check_attention_guard()` does it on demand.
    # Captured graphs (mmsa.Chains, SlideRunner, bench.py) cannot re-route themselves: their owner calls `check_attention_guard()`
    # after a replay (Chains.check_guard) and captures again when it reports a change.
    attention_guard = "sync"

    def _attn_policy(self):
        return getattr(self, "attention_precision", "auto")

    def _attn_mode(self, bp):
        """'f16' or 'b3' for this block's attention kernels.  `attention_precision`: 'f16' / 'b3' force one; 'auto'
        (default): the block's recorded mode -- 'f16' until its guard word has exceeded ATTN_F16_MAX_LOGIT once (check_attention_guard)."""
        pol = self._attn_policy()
        if pol in ("f16", "b3"):
            return pol
        return bp.get("amode") or "f16"

    def attention_guard_words(self):
        """The device tensor [depth] fp32 the attention kernels fold their largest |logit| into (None before the first forward): graph owners copy it to
        pinned memory behind a replay and hand the values to check_attention_guard(vals=...) (mmsa.Chains)."""
        return self._packed["attn_guard"] if self._packed is not None else None

    @torch.no_grad()
    def check_attention_guard(self, reroute=True, vals=None):
        """Read the per-block logit guard words (host sync; or take `vals`, a host copy of them made behind the passes in question) and fold them into
        the blocks' `max_logit`.  With `attention_precision = "auto"` a block that runs fp16 attention and has seen a logit above ATTN_F16_MAX_LOGIT is
        moved to fp16 hi/lo PAIR operands (f3 planes, clamped at +-65504: the attention kernels' q / k / v, the bias rows, the rel-pos tables and the block's
        four GEMM weights, repacked here) when `reroute`.  Returns the list of block indices that were (or, with reroute=False, would have to be) moved:
        non-empty means the outputs of the batches since the last check were computed on fp16 attention beyond the threshold -- run them again
        (forward() does by itself), and capture graphs again."""
        pk = self._packed
        if pk is None:
            return []
        if vals is None:
            vals = pk["attn_guard"].tolist()      # device -> host: waits for the work queued so far
        depth = self.cfg["depth"]
        clamped = max(float(vals[depth]) if len(vals) > depth else 0.0, float(pk.get("weights_clamped", 0.0)))
        if clamped > 0.0:
            # clamp watch: some kernel (or the pack: `weights_clamped`, sticky) converted a value beyond its operand format's range (h8 / h8c +-57344,
            # f3 +-65504) -- the planes hold the clamped value, the result is not the reference's
            if self.range_fallback and not self._wide():
                # wide-range state (see `range_fallback`): everything fp16-based moves to bf16 hi/lo pairs.  The pack is dropped -- the next forward
                # (forward() runs it at once, a graph owner's capture does) packs again -- and every block is reported as moved.  (reroute=False only
                # reports: the word stays set for the call that will act on it.)
                if reroute:
                    self._carry_modes = [(bp.get("amode"), bp.get("max_logit", 0.0)) for bp in pk["blocks"]]
                    self._wide_range = True
                    self._packed = None
                return list(range(depth))
            if len(vals) > depth and vals[depth] > 0.0:
                pk["attn_guard"][depth:].zero_()      # the activation part of the word (a later forward is judged on its own); a pack-time clamp stays flagged in the pack
            raise OperandRangeError(f"mmsa: a value of magnitude {clamped:.6g} or more (the GEMM's register epilogue reports the format's limit, not the value) was clamped on its way "
                                    "into fp16-based operand planes (h8 / h8c hold |x| <= 57344, f3 |x| <= 65504): the outputs since the last check are not the reference's.  "
                                    + ("The model is in its wide-range state already: what clamped is an operand of the attention kernels (q / k / v, a qkv bias or a rel-pos table "
                                       "entry beyond +-65504)." if self._wide() else "`range_fallback = True` (the default) re-routes instead of refusing."))
        auto = self._attn_policy() == "auto"
        moved = []
        sd_dev = None     # the float parameters on the device, built once for all the blocks that move in this call
        for bp, v in zip(pk["blocks"], vals):
            bp["max_logit"] = max(float(v), bp.get("max_logit", 0.0))
            if not auto or bp["qkv_bp16"] is None:
                continue
            if bp.get("amode") != "b3" and bp["max_logit"] > self.ATTN_F16_MAX_LOGIT:
                moved.append(bp["index"])
                if reroute:
                    bp["amode"] = "b3"
                    if bp["qkv"].fmt != self._pair_fmt():
                        # the projections around those logits must not lose them either: q and k from 2^-15.6 products turn a logit of 48 into
                        # an error of ~1e-3 before the exponential, and what proj / lin1 / lin2 lose reaches the NEXT block's q and k.  The
                        # whole block moves to hi/lo PAIR operands -- fp16 pairs since round 4 (f3 planes, 2^-22 per product; bf16 pairs before:
                        # 2^-17, a floor of 2^-17 x the logit) --, repacked here from the state dict.
                        dev = pk["attn_guard"].device
                        if sd_dev is None:
                            sd_dev = self._pack_state_dict(dev)
                        bp.update(self._block_gemm_planes(sd_dev, bp["index"], self._pair_fmt(), pk["fold_ln"], dev))
            elif bp.get("amode") is None:
                bp["amode"] = "f16"
        if moved and reroute and auto and self.inter_follow_blocks and "inter" in self._h8_sites():
            # the interactions of the moved blocks follow them onto pairs: their GEMM planes are packed inside _pack_impl's closures, so the pack is
            # dropped (the blocks' settled modes carry over) and the next forward -- forward() runs it at once, a graph owner's capture does -- packs again
            want = self._inter_pair_set() | {i for i in (self._interaction_of_block(b) for b in moved) if i is not None}
            if want != self._inter_pair_set():
                self._inter_pairs = want
                self._carry_modes = [(bp.get("amode"), bp.get("max_logit", 0.0)) for bp in pk["blocks"]]
                self._packed = None
        return moved

    def attention_modes(self):
        """[(mode, max |logit| seen)] per ViT block: 'f16' / 'b3' as the next forward will run it."""
        return [(self._attn_mode(bp), bp.get("max_logit", 0.0)) for bp in self._packed["blocks"]] if self._packed else []

    def calibrate_attention(self, x):
        """Show the model a batch ahead of deployment: an eager forward with the guard read back, whatever `attention_guard` says.  Returns
        the list of blocks that moved to bf16 hi/lo operands (graphs captured before must then be captured again)."""
        keep = self.attention_guard
        before = [bp.get("amode") for bp in self._packed["blocks"]] if self._packed else None
        self.attention_guard = "sync"
        try:
            self(x)
        finally:
            self.attention_guard = keep
        after = [bp.get("amode") for bp in self._packed["blocks"]]
        return [i for i, m in enumerate(after) if m == "b3" and (before is None or before[i] != "b3")]

    # ------------------------------------------------------------------ MSDeformAttn (ops/modules/ms_deform_attn.py:83-130)
    def _msda(self, ap, qn, fn, resid, out, ss, lsi, ref, B, Lq, S, L, colscale=None, stream_out=None, q_rn=None, f_rn=None):
        """MSDeformAttn.forward (ops/modules/ms_deform_attn.py:83-130) on normalised query / feature planes `qn` / `fn` -- or, with `q_rn` / `f_rn` =
        (mean_rstd, colsum), on the RAW tokens' planes with the LayerNorm applied by the projection's row-normalising epilogue.  `stream_out` =
        (planes, strip sums): the output projection also writes its result as operand planes + per-row strip sums (the producer half of a LayerNorm fold)."""
        ws, cfg = self._ws, self.cfg
        M, Pn = cfg["deform_num_heads"], cfg["n_points"]
        dv = ap["val"].n
        # `msda_value` (round 6; default "f32"): "f16" / "f16lo" hand the gather its values as H8 activation planes written by the value projection's epilogue --
        # half / three quarters of the bytes through the vector-memory pipe that bounds the gather (csrc/msda.hip msda_planes_kernel).  Not in the wide-range
        # state (the planes clamp at +-57344), not for head widths the planes kernel does not take
        vmode = getattr(self, "msda_value", "f32")
        if vmode in ("f16", "f16lo") and not self._wide() and (dv // M) % 8 == 0 and dv % 32 == 0:
            val = ws.planes("msda_valp", B * S, dv, fmt=ops.FMT_H8)
            ops.gemm(fn, ap["val"], bias=ap["val_b"], row_norm=f_rn, out_planes=val)
        else:
            val = ws.get("msda_val", B * S, dv)
            ops.gemm(fn, ap["val"], val, bias=ap["val_b"], row_norm=f_rn)
        raw = ws.get("msda_raw", B * Lq, ap["oa"].n)
        ops.gemm(qn, ap["oa"], raw, bias=ap["oa_b"], row_norm=q_rn)
        samp = ws.planes("msda_s", B * Lq, dv, fmt=ap["out"].fmt)
        ops.msda_fused(val, ss, lsi, raw, ref, None, B, S, M, dv // M, L, Lq, Pn, out_planes=samp, lo_bytes=vmode == "f16lo")
        if stream_out is not None:
            ops.gemm(samp, ap["out"], out, bias=ap["out_b"], resid=resid, colscale=colscale, out_planes=stream_out[0], rowstats_out=stream_out[1])
        else:
            ops.gemm(samp, ap["out"], out, bias=ap["out_b"], resid=resid, colscale=colscale)

    # Adapter-token LayerNorm fold (round 5, `_pack`: fold_adapter_ln).  `_cfold` = the state of one forward: "planes" hold the RAW operand planes of the
    # current c and "mr" its per-row (mean, rstd) -- valid from the first GEMM that writes c (an extractor's output projection or fc2) until the next one.
    def _c_fold_on(self, rows):
        """The fold runs when it was packed and the token rows fill whole 256-row GEMM tiles (the row-normalising epilogue on an fp32 output has no ragged form:
        other sizes run the same folded weights behind un-affine LayerNorm passes)."""
        return bool(self._packed.get("fold_adapter_ln")) and rows % 256 == 0

    def _c_buffers(self, rows, fmt):
        D = self.cfg["embed_dim"]
        return self._ws.planes("inj_fn", rows, D, fmt=fmt), self._ws.get("c_rs", rows, 2 * (D // 64)), self._ws.get("c_mr", rows, 2)

    def _c_produced(self, rows, fmt):
        """Called right after a GEMM wrote c together with its planes and strip sums: finalise the row statistics, mark the planes current."""
        cp, crs, cmr = self._c_buffers(rows, fmt)
        ops.rowstats_finalize(crs, rows, self.cfg["embed_dim"], 1e-6, cmr)
        self._cfold = dict(planes=cp, mr=cmr)

    def _injector(self, ip, x_in, x_out, c, geo, B, T, Nc):  # AM:525-542
        ws, D = self._ws, self.cfg["embed_dim"]
        qn = ws.planes("inj_qn", B * T, D, fmt=ip["attn"]["oa"].fmt)
        ops.layernorm(x_in, ip["qnw"], ip["qnb"], 1e-6, out_planes=qn)
        st = self._cfold
        f_rn = None
        if st is not None and "val_cs" in ip["attn"]:      # feat_norm inside value_proj: raw planes of c + its row statistics, from c's last producer
            fn, f_rn = st["planes"], (st["mr"], ip["attn"]["val_cs"])
        else:
            fn = ws.planes("inj_fn", B * Nc, D, fmt=ip["attn"]["val"].fmt)
            if ip["fold_c"]:   # chat = (c - mean) * rstd: feat_norm's affine part lives in value_proj's packed weight; the extractor reuses `fn`
                pk = self._packed
                ops.layernorm(c, pk["ln_one"], pk["ln_zero"], 1e-6, out_planes=fn)
            else:
                ops.layernorm(c, ip["fnw"], ip["fnb"], 1e-6, out_planes=fn)
        self._msda(ip["attn"], qn, fn, x_in, x_out, geo["ss1"], geo["lsi1"], geo["ref1"], B, T, Nc, 3, colscale=ip["gamma"],
                   stream_out=(self._stream_planes(B * T, self._packed["blocks"][ip["first_block"]]["qkv"].fmt)
                               if self._packed["fold_ln"] and B * T >= 128 else None), f_rn=f_rn)

    def _extractor(self, ep, c, x, geo, B, T, Nc, H, W, last=False):  # AM:490-511, ConvFFN AM:446-471
        """`last`: nothing reads c's operand planes after this extractor (the tail takes fp32): its fc2 writes none."""
        ws, D = self._ws, self.cfg["embed_dim"]
        pk = self._packed
        rows = B * Nc
        fold = self._c_fold_on(rows) and ep["fold_c"]
        cfmt = ep["attn"]["oa"].fmt
        st = self._cfold
        q_rn = None
        if fold and st is not None and "oa_cs" in ep["attn"]:   # query_norm inside the offsets / attention-weights projection
            qn, q_rn = st["planes"], (st["mr"], ep["attn"]["oa_cs"])
        else:
            qn = ws.planes("inj_fn", rows, D, fmt=cfmt)
            if not (ep["fold_c"] and ep["first"]):   # else: `inj_fn` still holds chat of this very c, written by the interaction's injector (nothing in between touches c or the buffer)
                if ep["fold_c"]:
                    ops.layernorm(c, pk["ln_one"], pk["ln_zero"], 1e-6, out_planes=qn)
                else:
                    ops.layernorm(c, ep["qnw"], ep["qnb"], 1e-6, out_planes=qn)
        fn = ws.planes("inj_qn", B * T, D, fmt=ep["attn"]["val"].fmt)
        ops.layernorm(x, ep["fnw"], ep["fnb"], 1e-6, out_planes=fn)
        produce = fold and (ep["cffn"] or not last)       # this extractor's output projection is followed by a consumer of LN(c)
        if produce:
            cp, crs, _ = self._c_buffers(rows, cfmt)
            self._cfold = None                             # (the planes are overwritten by the output projection below: q_rn's operand is read before, same stream)
            self._msda(ep["attn"], qn, fn, c, c, geo["ss2"], geo["lsi2"], geo["ref2"], B, Nc, T, 1, stream_out=(cp, crs), q_rn=q_rn)
            self._c_produced(rows, cfmt)
        else:
            self._cfold = None
            self._msda(ep["attn"], qn, fn, c, c, geo["ss2"], geo["lsi2"], geo["ref2"], B, Nc, T, 1, q_rn=q_rn)
        if not ep["cffn"]:   # AM:499-500
            return
        hid = ep["fc1"].n
        hp = ep["fc2"].kpad  # K of fc2 padded to a multiple of 32; pad columns stay zero
        h1 = ws.get("ffn_h1", rows, hid)
        h2f = ws.planes(f"ffn_h2_{hp}", rows, hp, zero=True, fmt=ep["fc2"].fmt)
        st = self._cfold
        if st is not None and "fc1_cs" in ep:             # ffn_norm inside fc1
            ops.gemm(st["planes"], ep["fc1"], h1, bias=ep["fc1_b"], row_norm=(st["mr"], ep["fc1_cs"]))
        else:
            qn = ws.planes("inj_fn", rows, D, fmt=ep["fc1"].fmt)
            if ep["fold_c"]:
                ops.layernorm(c, pk["ln_one"], pk["ln_zero"], 1e-6, out_planes=qn)
            else:
                ops.layernorm(c, ep["ffw"], ep["ffb"], 1e-6, out_planes=qn)
            ops.gemm(qn, ep["fc1"], h1, bias=ep["fc1_b"])
        off = 0
        for (hh, wwd) in ((H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)):  # AM:462-470 token split 16n/4n/n
            ops.dwconv(h1[off:], ep["dw"], ep["dw_b"], None, B, hh, wwd, 3, act="gelu",
                       xstride_b=Nc * hid, out_planes=h2f.rows(off), pstride_b=Nc * 2 * hp)
            off += hh * wwd
        self._cfold = None
        if fold and not last:
            cp, crs, _ = self._c_buffers(rows, cfmt)
            ops.gemm(h2f, ep["fc2"], c, bias=ep["fc2_b"], resid=c, out_planes=cp, rowstats_out=crs)
            self._c_produced(rows, cfmt)
        else:
            ops.gemm(h2f, ep["fc2"], c, bias=ep["fc2_b"], resid=c)

    # ------------------------------------------------------------------ spatial prior module (AM:929-964)
    def _spm(self, x, B, H, W, c1_out, cbuf, Nc, join=True):
        pk, ws = self._packed, self._ws
        D = self.cfg["embed_dim"]
        chans = self.channels
        sizes = [(H // 4, W // 4), (H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
        tcat = [ws.get(f"tcat{i}", B * sizes[i][0] * sizes[i][1], 2 * chans[i]) for i in range(4)]
        # the same maps as interleaved planes (A operand of the neck's 1x1 convs) when the per-modality width is a k-block multiple
        tcat_p = [ws.planes(f"tcat{i}", B * sizes[i][0] * sizes[i][1], 2 * chans[i]) if chans[i] % 32 == 0 else None for i in range(4)]
        # The ConvNeXt chain and the four neck levels are chains of mostly small launches (M = 8192 rows at 1/16 resolution
        # fills a fraction of the CUs per GEMM), so the neck levels run on separate HIP streams (fork/join with events;
        # captured as parallel branches of the HIP graph) with private workspaces.  Neck level i needs only the stage-i
        # outputs, so it starts as soon as they are written: the heavy 1/4-resolution level runs underneath ConvNeXt
        # stages 1..3 instead of after them.
        main = torch.cuda.current_stream()
        multi = getattr(self, "multistream", True)
        if multi:
            if not hasattr(self, "_sides"):
                self._sides = {}
            skey = (getattr(self._ws, "tag", ""), x.device)   # concurrent chains (mmsa.chains) fork onto their own side streams
            if skey not in self._sides:
                self._sides[skey] = [torch.cuda.Stream(device=x.device) for _ in range(4)]
            s_neck = self._sides[skey]
        else:   # profiling aid: everything on the current stream
            s_neck = [main] * 4
        # --- TwinConvNeXt (TC:445-476): the rgb and the auxiliary stream are ONE chain of batched kernels (batch index =
        #     stream, per-batch weights): every launch carries twice the work of a per-stream launch.  As two concurrent
        #     HIP streams the chains took 8.5 ms against 5.5 ms for one alone -- the persistent GEMM grids of one stream
        #     hold the LDS of every CU, so the other stream's kernels queue behind them.
        ev_x = []
        self._twin_batched(x, B, sizes, tcat, ev_x, tcat_p)
        if self._taps is not None:
            for i in range(4):
                self._taps[f"twin{i}"] = tcat[i].clone()
        # --- RoadFormer2Neck (AM:364-394) + fc1..4 (AM:947-956): one stream per level, gated on that level's two inputs
        offs = [0, 0, sizes[1][0] * sizes[1][1], sizes[1][0] * sizes[1][1] + sizes[2][0] * sizes[2][1]]
        joins = []
        for i in range(4):
            sn = s_neck[i]
            sn.wait_event(ev_x[i])
            with torch.cuda.stream(sn):
                if i == 0:
                    self._neck_level(0, pk["neck"][0], tcat[0], B, sizes[0][0], sizes[0][1], chans[0], c1_out, 0, tcat_p[0])
                else:
                    self._neck_level(i, pk["neck"][i], tcat[i], B, sizes[i][0], sizes[i][1], chans[i], cbuf[offs[i]:], Nc * D, tcat_p[i])
                e = torch.cuda.Event()
                e.record(sn)
                joins.append(e)
        # c1 (level 0, the heaviest) is consumed only by the tail (BK:316-337): its stream is joined there, so it runs
        # underneath the ViT blocks; the injectors need levels 1..3
        if not join:
            return joins   # pipelined mode: the caller's stream joins every level
        for e in joins[1:]:
            main.wait_event(e)
        return joins[0]

    def _twin_batched(self, x, B, sizes, tcat, stage_events, tcat_p=None):
        """Both ConvNeXt streams (TC:451-472) as one chain: activations stacked [2 (stream), B*h*w, c], weights [2, ...];
        stage outputs are channel-concatenated into tcat[i] (TC:466-472) by the grouped out-norm LayerNorm."""
        pk, ws = self._packed, self._ws
        chans = self.channels
        st = pk["twin2"]
        t = "cnb_"
        h0, w0 = sizes[0]
        P0 = B * h0 * w0
        kp = st["stem"].kpad
        a = ws.get(t + "stem_a", 2 * P0, kp)
        for si in range(2):
            ops.im2col_nchw(x, 3 * si, 3, 4, a[si * P0:(si + 1) * P0])
        t0 = ws.get(t + "tmp", 2 * P0, chans[0])
        ops.gemm(a, st["stem"], t0, bias=st["stem_b"], batch=2, m=P0, stride_a=P0 * kp, stride_w=st["stem"].n * 2 * kp,
                 stride_bias=chans[0], stride_c=P0 * chans[0])
        cur = ws.get(t + "cur0", 2 * P0, chans[0])
        ops.layernorm(t0, st["stem_nw"], st["stem_nb"], 1e-6, cur, group_rows=P0, w_gstride=chans[0])
        for i in range(4):
            hh, wwd = sizes[i]
            P = B * hh * wwd
            c = chans[i]
            if i >= 1:
                ds = st["ds"][i - 1]
                cp = chans[i - 1]
                Pp = B * sizes[i - 1][0] * sizes[i - 1][1]
                pa = ws.planes(t + "patch", 2 * P, 4 * cp, fmt=ds["w"].fmt)
                ops.layernorm(cur, ds["nw"], ds["nb"], 1e-6, out_planes=pa, patchify=(sizes[i - 1][0], sizes[i - 1][1]),
                              group_rows=Pp, w_gstride=cp)
                cur = ws.get(t + f"cur{i}", 2 * P, c)
                ops.gemm(pa, ds["w"], cur, bias=ds["b"], batch=2, m=P, stride_a=P * 2 * pa.kpad, stride_w=ds["w"].n * 2 * ds["w"].kpad,
                         stride_bias=c, stride_c=P * c)
            d = ws.get(t + "tmp", 2 * P, c)
            wf = st["stages"][i][0]["pw1"].fmt          # operand format of this stage's pointwise convs (bf16 hi/lo unless "cnx" is an h8 site)
            n = ws.planes(t + "n", 2 * P, c, fmt=wf)
            hbuf = ws.planes(t + "h", 2 * P, 4 * c, fmt=st["stages"][i][0]["pw2"].fmt)   # (pointwise_conv2's operand format: f3 like pw1, or h8c with the "cnx2p2" site)
            # narrow stages (C = 96 / 192): the pointwise pair as ONE kernel that keeps the 4C hidden tensor in LDS (csrc/mlp_fused.hip)
            fuse_mlp = ops.convnext_mlp_fused_supported(c) and bool(getattr(self, "fuse_convnext_mlp", True))
            fold = "pw1f" in st["stages"][i][0] and not fuse_mlp and hh % 8 == 0 and wwd % 8 == 0 and P >= 128
            if fold:
                rs = ws.get(t + "rs", 2 * P, 2 * (c // 64))
                mr = ws.get(t + "mr", 2 * P, 2)
            for blk in st["stages"][i]:  # ConvNeXtBlock TC:98-132
                if fold:
                    # LayerNorm folded (like the ViT blocks', LAB_NOTES.md 4.2): the depthwise conv writes its RAW output as planes + per pixel and
                    # 64-channel chunk (sum, sum of squares); pointwise_conv1 runs on W o w and normalises in its epilogue
                    ops.dwconv(cur, blk["dw"], blk["dw_b"], None, 2 * B, hh, wwd, 7, imgs_per_group=B, out_planes=n, rowstats_out=rs)
                    ops.rowstats_finalize(rs, 2 * P, c, 1e-6, mr)
                    ops.gemm(n, blk["pw1f"], bias=blk["pw1_bf"], act="gelu", out_planes=hbuf, row_norm=(mr, blk["pw1_cs"]), batch=2, m=P,
                             stride_a=P * 2 * n.kpad, stride_w=blk["pw1f"].n * 2 * blk["pw1f"].kpad, stride_bias=4 * c, stride_cp=P * 2 * hbuf.kpad)
                    ops.gemm(hbuf, blk["pw2"], cur, bias=blk["pw2_b"], colscale=blk["gamma"], resid=cur, batch=2, m=P,
                             stride_a=P * 2 * hbuf.kpad, stride_w=blk["pw2"].n * 2 * blk["pw2"].kpad, stride_bias=c,
                             stride_r=P * c, stride_c=P * c)
                    continue
                ops.dwconv(cur, blk["dw"], blk["dw_b"], d, 2 * B, hh, wwd, 7, imgs_per_group=B)
                ops.layernorm(d, blk["nw"], blk["nb"], 1e-6, out_planes=n, group_rows=P, w_gstride=c)
                if fuse_mlp:
                    ops.convnext_mlp_fused(n, blk["pw1"], blk["pw2"], blk["pw1_b"], blk["pw2_b"], blk["gamma"], cur, P, batch=2,
                                           stride_a=P * 2 * n.kpad, stride_w1=blk["pw1"].n * 2 * blk["pw1"].kpad,
                                           stride_w2=blk["pw2"].n * 2 * blk["pw2"].kpad, stride_x=P * c)
                    continue
                ops.gemm(n, blk["pw1"], bias=blk["pw1_b"], act="gelu", out_planes=hbuf, batch=2, m=P, stride_a=P * 2 * n.kpad,
                         stride_w=blk["pw1"].n * 2 * blk["pw1"].kpad, stride_bias=4 * c, stride_cp=hbuf.batch_stride(P))
                ops.gemm(hbuf, blk["pw2"], cur, bias=blk["pw2_b"], colscale=blk["gamma"], resid=cur, batch=2, m=P,
                         stride_a=hbuf.batch_stride(P), stride_w=blk["pw2"].batch_stride(blk["pw2"].n), stride_bias=c,
                         stride_r=P * c, stride_c=P * c)
            nw, nb = st["out_norm"][i]
            ops.layernorm(cur, nw, nb, 1e-6, tcat[i], group_rows=P, w_gstride=c, y_gcol=c, y_wrap=True,
                          out_planes=tcat_p[i] if tcat_p is not None else None)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            stage_events.append(ev)

    def _neck_level(self, level, lv, t, B, h, w, c, out, out_stride_b, tp=None):
        ws = _Tagged(self._ws, f"nk{level}_")
        C = 2 * c
        HW = h * w
        P = B * HW
        gcat = ws.get("nk_g", P, C)
        lcat = ws.get("nk_l", P, C)
        # When the per-modality width is a k-block multiple (every shipped config), the 1x1-conv operands travel as
        # interleaved planes written by their producers (LayerNorm, GEMM epilogue, depthwise conv, ca_apply), so all of the
        # neck's GEMMs run on the LDS-DMA kernel instead of the fp32-A one.
        pl_ok = tp is not None and c % 32 == 0
        gcat_p = ws.planes("nk_gp", P, C) if pl_ok else None
        lcat_p = ws.planes("nk_lp", P, C) if pl_ok else None
        # the level's three double-precision column-statistics buffers (atomically summed row slices) are slices of ONE block zeroed by
        # ONE memset at the head of the level -- a memset node per call was 24 per forward (0.1-0.2 ms per step; the Grams need none)
        n_st, n_g, n_st2 = B * 3 * 3 * c, B * c * c, B * 3 * C
        acc_blk = ws.get("nk_acc", 1, 2 * (n_st + n_g) + n_g + n_st2, dtype=torch.float64)
        ops.zero_(acc_blk[0, :2 * n_st + n_st2])
        gram_scr = ws.get("nk_gscr", 1, (ops.gram_tn_scratch_bytes(B, HW, c) + 3) // 4)
        acc_off = [0, 2 * n_st + n_st2]

        def acc_buf(rows, cols, gram=False):   # statistics from the zeroed head of the block, Grams behind it
            o = acc_off[int(gram)]
            acc_off[int(gram)] = o + rows * cols
            return acc_blk[0, o:o + rows * cols].view(rows, cols)
        for m in range(2):
            X = t[:, m * c:(m + 1) * c]
            # GFE (AM:133-145, 75-109): x + LN(x) + proj(softmax(norm(q) norm(k)^T * temp) v) * scale2
            gp = lv["gfe"][m]
            y = ws.get("nk_y", P, c)
            s = ws.get("nk_s", P, c)
            ops.layernorm(X, gp["nw"], gp["nb"], 1e-5, y, out2=s)
            q1 = ws.get("nk_q1", P, 3 * c)
            q2 = ws.get("nk_q2", P, 3 * c)
            ops.gconv(y, gp["q1"], None, q1, B, h, w, 32, c // 32, 3 * c // 32, 1)
            ops.gconv(q1, gp["q2"], None, q2, B, h, w, 32, 3 * c // 32, 3 * c // 32, 3)
            st = acc_buf(B * 3, 3 * c)
            ops.colstats(q2, HW * 3 * c, B, HW, st, out_is_zero=True)
            g = acc_buf(B * c, c, gram=True)
            ops.gram_tn(q2[:, :c], q2[:, c:2 * c], HW * 3 * c, g, B, HW, nblk=8, scratch=gram_scr)
            cp = ops.pad32(c)
            pl = ops.Planes(ws.get(f"nk_pp{c}", B * c, 2 * cp, dtype=torch.int16, zero=True), c, c, cp)
            base = st.data_ptr()
            ops.chanattn_build(g, base + 8 * (3 * c), 9 * c, base + 8 * (3 * c + c), 9 * c, gp["temp"], gp["proj"], pl, B, c, 8)
            ops.gemm(q2[:, 2 * c:], pl, gcat[:, m * c:(m + 1) * c], alpha=gp["scale2"], resid=s, batch=B, m=HW,
                     stride_a=HW * 3 * c, stride_w=c * 2 * cp, stride_r=HW * c, stride_c=HW * C,
                     out_planes=gcat_p.cols(m * c, (m + 1) * c) if pl_ok else None, stride_cp=HW * 2 * C)
            # MobileNetV2 (AM:281-295)
            lp = lv["loc"][m]
            h1 = ws.get("nk_q1", P, 2 * c)
            h2 = ws.get("nk_q2", P, 2 * c)
            if pl_ok:
                ops.gemm(tp.cols(m * c, (m + 1) * c), lp["w1"], h1, act="relu6")
                h2p = ws.planes("nk_h2p", P, 2 * c)
                ops.dwconv(h1, lp["dw"], None, None, B, h, w, 3, act="relu6", out_planes=h2p)
                ops.gemm(h2p, lp["w3"], alpha=lp["scale"], resid=X, out_planes=lcat_p.cols(m * c, (m + 1) * c))
            else:
                ops.gemm(X, lp["w1"], h1, act="relu6")
                ops.dwconv(h1, lp["dw"], None, h2, B, h, w, 3, act="relu6")
                ops.gemm(h2, lp["w3"], lcat[:, m * c:(m + 1) * c], alpha=lp["scale"], resid=X)
        # GFFM (AM:242-267)
        e = acc_buf(B * c, c, gram=True)
        ops.gram_tn(gcat[:, :c], gcat[:, c:], HW * C, e, B, HW, nblk=1, scratch=gram_scr)
        cp = ops.pad32(c)
        px = ops.Planes(ws.get(f"nk_pp{c}", B * c, 2 * cp, dtype=torch.int16, zero=True), c, c, cp)
        py = ops.Planes(ws.get(f"nk_pp2{c}", B * c, 2 * cp, dtype=torch.int16, zero=True), c, c, cp)
        ops.gffm_build(e, px, py, B, c)
        fbuf = ws.get("nk_f", P, C)
        ops.gemm(gcat_p.cols(c, C) if pl_ok else gcat[:, c:], px, fbuf[:, :c], alpha=lv["gx"], resid=gcat[:, :c], batch=B, m=HW,
                 stride_a=HW * C * (2 if pl_ok else 1), stride_w=c * 2 * cp, stride_r=HW * C, stride_c=HW * C)
        ops.gemm(gcat_p.cols(0, c) if pl_ok else gcat[:, :c], py, fbuf[:, c:], alpha=lv["gy"], resid=gcat[:, c:], batch=B, m=HW,
                 stride_a=HW * C * (2 if pl_ok else 1), stride_w=c * 2 * cp, stride_r=HW * C, stride_c=HW * C)
        # LayerNorm over H*W (AM:265) + FFRM (AM:158-162), one apply pass
        st = acc_buf(B * 3, C)
        ops.colstats(fbuf, HW * C, B, HW, st, wrow=lv["lnw"], out_is_zero=True)
        mean = ws.get("nk_mean", B, C)
        rstd = ws.get("nk_rstd", B, C)
        mult = ws.get("nk_mult", B, C)
        ops.ffrm_finalize(st, B, HW, C, lv["lnw_mean"], lv["lnb_mean"], lv["ffrm_w"], lv["gn_w"], lv["gn_b"], mean, rstd, mult,
                          ws.get("nk_ffrm", 2 * B, C))
        fn = ws.get("nk_g", P, C)  # gcat is dead now
        ops.lnhw_apply(fbuf, mean, rstd, mult, lv["lnw"], lv["lnb"], fn, B, HW)
        # gated MLP on the local branch (AM:127-132) and Scale2 (AM:279-280)
        hm = ws.get("nk_hm", P, 2 * C)
        ops.gemm(lcat_p if pl_ok else lcat, lv["mlp_in"], hm)
        hg = ws.planes("nk_hg", P, C)   # dw 3x3 pair conv + chunk + gelu gate fused, emitted as planes for the GEMM
        ops.dwpair_gate(hm, lv["mlp_dw"], None, B, h, w, C, out_planes=hg)
        z = ws.get("nk_f", P, C)  # fbuf is dead now
        ops.gemm(hg, lv["mlp_out"], z, alpha=lv["s2"], resid=fn, beta=lv["s1"])
        # CoordinateAttention (AM:187-201) + residual (AM:218-221)
        pooled = ws.get("nk_pool", B * (h + w), C)
        ops.pool_hw(z, pooled, B, h, w)
        y1 = ws.get(f"nk_y1_{lv['mip_pad']}", B * (h + w), lv["mip_pad"], zero=True)
        ops.gemm(pooled, lv["ca1"], y1, bias=lv["ca1_b"], act="hswish")
        att = ws.get("nk_att", B * (h + w), C)
        ops.gemm(y1, lv["cah"], att, bias=lv["cah_b"], act="sigmoid", batch=B, m=h,
                 stride_a=(h + w) * lv["mip_pad"], stride_c=(h + w) * C)
        ops.gemm(y1[h:], lv["caw"], att[h:], bias=lv["caw_b"], act="sigmoid", batch=B, m=w,
                 stride_a=(h + w) * lv["mip_pad"], stride_c=(h + w) * C)
        if pl_ok:
            zo = ws.planes("nk_zop", P, C)
            ops.ca_apply(z, att, None, B, h, w, out_planes=zo)
            sa = HW * 2 * C
        else:
            zo = ws.get("nk_g", P, C)  # fn is dead now
            ops.ca_apply(z, att, zo, B, h, w)
            sa = HW * C
        if self._taps is not None:
            self._taps[f"fuse{level}"] = ops.planes_to_float(zo, cols=C) if pl_ok else zo[:, :C].clone()
        # fc_i (AM:947-956) straight into c1 / the c2|c3|c4 token buffer (level embed folded into the bias)
        if out_stride_b == 0:
            ops.gemm(zo, lv["fc"], out, bias=lv["fc_b"])
        else:
            ops.gemm(zo, lv["fc"], out, bias=lv["fc_b"], batch=B, m=HW, stride_a=sa, stride_c=out_stride_b)


class SAMAdapterbimodalMixModNewInTwinConvNEWwithcp(SAMAdapterbimodalMixModNewInTwinConvNEW):
    """Second registered name (backbones/__init__.py:3-9): identical inference math, extra training-only kwargs."""


# ---------------------------------------------------------------------- pack-time helpers (host/torch, one-time)
def _ref_points(shapes):
    """get_reference_points (AM:397-409): per-query (x, y) in [0,1], levels concatenated -> [Lq, 2] fp32."""
    refs = []
    for (h, w) in shapes:
        ys = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32) / h
        xs = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32) / w
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        refs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    return torch.cat(refs, 0).contiguous()


def rel_pos_index(q_size, k_size):
    """Integer gather table of get_rel_pos (IE:579-584): float coords then .long() -- kept bit-exact."""
    q = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    return ((q - k) + (k_size - 1) * max(q_size / k_size, 1.0)).long()


def _linear_resize_rows(tab, L):
    """F.interpolate(mode='linear', align_corners=False) along dim 0 of a [Lin, C] table (IE:568-575)."""
    Lin = tab.shape[0]
    scale = Lin / L
    pos = (torch.arange(L, dtype=torch.float32, device=tab.device) + 0.5) * scale - 0.5
    pos = pos.clamp(min=0)
    i0 = pos.floor().long().clamp(max=Lin - 1)
    i1 = (i0 + 1).clamp(max=Lin - 1)
    lam = (pos - i0.float()).unsqueeze(1)
    return tab[i0] * (1 - lam) + tab[i1] * lam


def _rel_table(size, rel_pos):
    """get_rel_pos(size, size, rel_pos) -> gathered [size, size, hd] table (IE:554-584)."""
    L = 2 * size - 1
    tab = rel_pos if rel_pos.shape[0] == L else _linear_resize_rows(rel_pos, L)
    return tab[rel_pos_index(size, size).to(tab.device)].contiguous()


def _cubic_w(t, A=-0.75):
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return torch.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], -1)


def _bicubic_matrix(n_in, n_out, device):
    scale = n_in / n_out
    x = (torch.arange(n_out, dtype=torch.float32, device=device) + 0.5) * scale - 0.5
    ix = x.floor()
    w = _cubic_w(x - ix)  # [n_out, 4]
    m = torch.zeros(n_out, n_in, dtype=torch.float32, device=device)
    for k in range(4):
        idx = (ix.long() - 1 + k).clamp(0, n_in - 1)
        m.scatter_add_(1, idx[:, None], w[:, k:k + 1])
    return m


def _bicubic_resize(src_hwc, H, W):
    """F.interpolate(mode='bicubic', align_corners=False) of a [h, w, C] map (BK:136-143); identity when sizes match."""
    h, w, _ = src_hwc.shape
    if (h, w) == (H, W):
        return src_hwc.contiguous()
    my = _bicubic_matrix(h, H, src_hwc.device)
    mx = _bicubic_matrix(w, W, src_hwc.device)
    return torch.einsum("Hh,hwc,Ww->HWc", my, src_hwc, mx).contiguous()
