"""Checkpoint ingestion of the drop-in backbone.

  load_pretrained(model, path)   -- what ImageEncoderViT.init_weights does (IE:305-315): the reference's non-strict
                                    `mmcv_custom.load_checkpoint` (segmentation/mmcv_custom/checkpoint.py:319-514) restated for the
                                    keys a SAM image-encoder checkpoint has.
  convert_sam_release(sd)        -- segmentation/tools/SAM_checkpoint_convert.py:15-33: keep 'image_encoder' keys, drop the neck,
                                    strip the 'image_encoder.' prefix.
  save_packed / load_packed      -- the backbone's one-time weight pre-pack (bf16 hi/lo planes, folded norms, re-laid-out conv
                                    weights) written once next to the plain state dict (torch zip-pickle, Appendix A.3 keys), so
                                    that a serving process does not re-split 456 M parameters at its first forward.

Pinned by tests/test_host_cpu.py against tests/golden/sam_ckpt.npz, which the reference's own loader and converter produced on
seeded checkpoints (tools/oracle/make_golden.py::gen_sam_ckpt)."""
import hashlib
import os
import warnings

import torch

from . import ops

PACK_FORMAT = 11   # 11 (round 6): extractor dicts carry `first`, fc1 of every shared-norm extractor the folded ffn_norm, attn_guard has depth + 1 words (the clamp watch), the offsets / attention-weights projection is padded to 128 columns, settings carry fold_adapter_ln and the wide-range state, the pack its `weights_clamped` flag; 10: the attention path's hi/lo planes (bias rows, rel-pos tables, fallback block weights) are fp16 pairs; 9: the ConvNeXt planes' format (fp16 hi/lo pairs) among the settings; 8: byte-exact digests (sha1) instead of floating-point sums; the blocks' attention modes / largest logits travel with the planes; 7: ConvNeXt LayerNorm fold (pw1f / pw1_cs / pw1_bf, setting fold_cnx_ln); 6: plane checksum + pack-time settings in the header, both attention table formats (relp / relp16, qkv_bp_b3); 3: planes carry their operand format (bf16 hi/lo or h8); 4: LayerNorm affine parts folded into the adapter projections (share_c_norm); 5: planes carry `split` (qkv bias rows: v columns as h8 planes)


def unwrap_state_dict(ck):
    """checkpoint.py:343-360: 'state_dict' / 'model' / 'module' containers, then a leading 'module.' (DataParallel) prefix on the
    FIRST key strips 7 characters from every key, then -- when the first key in sorted order starts with 'encoder' (MoBY) -- only
    'encoder.' keys are kept, with that substring removed."""
    if not isinstance(ck, dict):
        raise RuntimeError("No state_dict found in checkpoint file")        # checkpoint.py:339-341
    sd = ck
    for key in ("state_dict", "model", "module"):
        if key in ck:
            sd = ck[key]
            break
    if list(sd.keys())[0].startswith("module."):
        sd = {k[7:]: v for k, v in sd.items()}
    if sorted(list(sd.keys()))[0].startswith("encoder"):
        sd = {k.replace("encoder.", ""): v for k, v in sd.items() if k.startswith("encoder.")}
    return dict(sd)


def _resize_pos_embed_like_reference(sd, model):
    """checkpoint.py:447-470 as it acts on THIS model (4-D SAM pos_embed [1, Hp, Wp, C], `patch_embed.num_patches` =
    (img/patch)^2): num_extra_tokens = pos_embed.shape[-2] - num_patches, orig_size = int(sqrt(ckpt.shape[-2] - num_extra_tokens)),
    new_size = int(sqrt(num_patches)).  For a checkpoint grid equal to the model's the two sizes coincide and nothing happens (the
    resize to the input resolution is done at forward time, BK:136-143); any other grid makes the reference's reshape fail, which
    is reported here instead of silently loading something else."""
    if "pos_embed" not in sd:
        return
    pe = sd["pos_embed"]
    num_patches = (model.cfg["img_size"] // model.cfg["patch_size"]) ** 2
    num_extra = model.pos_embed.shape[-2] - num_patches
    orig_size = int((pe.shape[-2] - num_extra) ** 0.5)
    new_size = int(num_patches ** 0.5)
    if orig_size != new_size:
        raise RuntimeError(f"pos_embed of the checkpoint ({tuple(pe.shape)}) does not match the model's ({tuple(model.pos_embed.shape)}): "
                           "the reference's loader (mmcv_custom/checkpoint.py:460-470) cannot resize a 4-D SAM position embedding "
                           "either; build the model with the checkpoint's pretrained_size")


def load_pretrained(model, path, map_location="cpu"):
    """Non-strict load with the reference's semantics: unexpected and missing keys are tolerated, a key whose shape differs from
    the model's is SKIPPED with a warning (mmcv's load_state_dict collects the size-mismatch message of
    nn.Module._load_from_state_dict and only warns, checkpoint.py:44-113).  Returns (loaded, skipped, unexpected) key lists."""
    sd = unwrap_state_dict(torch.load(path, map_location=map_location))
    _resize_pos_embed_like_reference(sd, model)
    own = model.state_dict()
    good, skipped, unexpected = {}, [], []
    for k, v in sd.items():
        if k not in own:
            unexpected.append(k)
        elif tuple(own[k].shape) != tuple(v.shape):
            skipped.append(k)
        else:
            good[k] = v
    if skipped:
        warnings.warn("mmsa: size mismatch, not loaded (the model keeps its initialisation): " + ", ".join(skipped))
    model.load_state_dict(good, strict=False)
    return sorted(good), skipped, unexpected


def convert_sam_release(state_dict):
    """tools/SAM_checkpoint_convert.py:15-33 on an in-memory SAM release state dict."""
    kept = {k: v for k, v in state_dict.items() if "image_encoder" in k}
    kept = {k: v for k, v in kept.items() if "neck" not in k}
    return {k.replace("image_encoder.", ""): v for k, v in kept.items()}


# ---------------------------------------------------------------------------------------------- packed form
def _enc(o):
    if isinstance(o, ops.Planes):
        full = getattr(o, "full", None)
        return {"__planes__": True, "p": (full if full is not None else o.p).cpu(), "n": o.n, "k": o.k, "kpad": o.kpad, "stacked": full is not None,
                "fmt": o.fmt, "weight": o.weight, "split": o.split}
    if isinstance(o, torch.Tensor):
        return o.cpu()
    if isinstance(o, dict):
        return {k: _enc(v) for k, v in o.items() if k not in ("geom", "dev")}
    if isinstance(o, (list, tuple)):
        return [_enc(v) for v in o]
    return o


def _dec(o, dev):
    if isinstance(o, dict) and o.get("__planes__"):
        buf = o["p"].to(dev)
        if o["stacked"]:
            rows_ = ops.planes_shape(o["n"], o["k"], o.get("fmt", ops.FMT_B3))[0]      # (h8c planes hold row pairs)
            pl = ops.Planes(buf[:rows_], o["n"], o["k"], o["kpad"], o.get("fmt", ops.FMT_B3), o.get("weight", False), o.get("split", 0))
            pl.full = buf
            return pl
        return ops.Planes(buf, o["n"], o["k"], o["kpad"], o.get("fmt", ops.FMT_B3), o.get("weight", False), o.get("split", 0))
    if isinstance(o, torch.Tensor):
        return o.to(dev)
    if isinstance(o, dict):
        return {k: _dec(v, dev) for k, v in o.items()}
    if isinstance(o, list):
        return [_dec(v, dev) for v in o]
    return o


def _digest_tensor(h, t):
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.bfloat16:
        t = t.view(torch.int16)
    h.update(str((tuple(t.shape), str(t.dtype))).encode())
    h.update(t.numpy().tobytes())


def _fingerprint(sd):
    """Identity of a state dict: key list + a sha1 over every tensor's shape, dtype and BYTES in key order (exact on every host: the
    floating-point sums used before depended on the reduction order of the torch build that computed them)."""
    h = hashlib.sha1()
    for k, v in sd.items():
        h.update(k.encode())
        _digest_tensor(h, v)
    return [list(sd.keys()), h.hexdigest()]


def _packed_checksum(enc):
    """sha1 over every tensor of the ENCODED packed tree, in traversal order (the plane buffers as stored): what load_packed verifies
    before it trusts the planes instead of repacking."""
    h, n = hashlib.sha1(), [0]

    def walk(o):
        if isinstance(o, torch.Tensor):
            n[0] += 1
            _digest_tensor(h, o)
        elif isinstance(o, dict):
            for k, v in o.items():
                h.update(str(k).encode())
                walk(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
    walk(enc)
    return [n[0], h.hexdigest()]


def save_packed(model, path, device="cuda"):
    """Pack `model`'s current weights on `device` (runs the HIP split kernels once) and write {state_dict, packed, fingerprint, the
    checksum of the packed planes, the pack-time settings the planes depend on}."""
    dev = torch.device(device)
    with torch.cuda.device(dev):
        pk = model._pack(dev)
        # the per-block attention precision the model has settled on (backbone.check_attention_guard) travels with the planes: a block
        # that was moved to bf16 hi/lo operands is packed that way, and every block keeps the largest logit it has seen
        old = getattr(model, "_packed", None)
        if old is not None:
            sd_dev = model._pack_state_dict(dev)
            for bp, bo in zip(pk["blocks"], old["blocks"]):
                for k in ("amode", "max_logit"):
                    if k in bo:
                        bp[k] = bo[k]
                if bo.get("amode") == "b3" and bp["qkv"].fmt != model._pair_fmt():
                    bp.update(model._block_gemm_planes(sd_dev, bp["index"], model._pair_fmt(), pk["fold_ln"], dev))
        torch.cuda.synchronize(dev)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    enc = _enc(pk)
    torch.save({"format": PACK_FORMAT, "cfg": model.cfg, "state_dict": sd, "packed": enc, "fingerprint": _fingerprint(sd),
                "packed_checksum": _packed_checksum(enc),
                "settings": {"h8_sites": list(model._h8_sites()), "h8c": bool(pk.get("h8c", False)), "share_c_norm": bool(pk.get("share_c_norm", True)), "fold_ln": bool(pk.get("fold_ln", False)),
                             "fold_cnx_ln": bool(pk.get("fold_cnx_ln", False)), "cnx_f16": bool(pk.get("cnx_f16", False)),
                             "fold_adapter_ln": bool(pk.get("fold_adapter_ln", False)), "wide": bool(pk.get("wide", False)),
                             "inter_pairs": list(pk.get("inter_pairs", []))}}, path)


def load_packed(model, path, device="cuda"):
    """Load a file written by save_packed: the plain state dict (strict) AND the packed planes, so the first forward skips _pack.
    Refuses a file packed for another architecture, with other pack-time settings (operand-format sites, shared c-norm folding) than
    the model's current ones, whose state dict or plane buffers fail their checksums, or whose planes are not the split of the weights
    it carries (spot check: two weight matrices decoded back from their planes)."""
    blob = torch.load(path, map_location="cpu")
    if not isinstance(blob, dict) or blob.get("format") != PACK_FORMAT:
        raise RuntimeError(f"{path}: not an mmsa packed checkpoint (format {PACK_FORMAT})")
    if blob["cfg"] != model.cfg:
        raise RuntimeError(f"{path}: packed for a different architecture")
    # the wide-range state (backbone.range_fallback) is a property of the WEIGHTS the file carries: it travels with them.  (After load_state_dict below --
    # its post hook resets the state -- it is set from the file again.)
    wide = bool((blob.get("settings") or {}).get("wide", False))
    inter_pairs = set((blob.get("settings") or {}).get("inter_pairs", []))   # interactions that followed their blocks onto pairs (backbone.inter_follow_blocks): settled state, like the blocks' modes
    model._wide_range = wide
    model._inter_pairs = inter_pairs
    want = {"h8_sites": list(model._h8_sites()), "h8c": bool(model._h8c_wanted()),
            "share_c_norm": bool(getattr(model, "share_c_norm", True))}
    want["fold_ln"] = bool(model._fold_ln_wanted())
    want["fold_cnx_ln"] = bool(getattr(model, "fold_convnext_ln", False))
    want["cnx_f16"] = bool(model._cnx_f16_wanted())
    want["wide"] = wide
    want["inter_pairs"] = sorted(inter_pairs)
    want["fold_adapter_ln"] = bool(model._fold_adapter_ln_wanted())   # (a pack-time setting that drives the run-time path: ADVICE r05)
    if blob.get("settings") != want:
        raise RuntimeError(f"{path}: packed with settings {blob.get('settings')}, the model runs {want}: repack")
    if blob["fingerprint"] != _fingerprint(blob["state_dict"]):
        raise RuntimeError(f"{path}: the state dict fails its checksum")
    if blob.get("packed_checksum") != _packed_checksum(blob["packed"]):
        raise RuntimeError(f"{path}: state dict and packed planes do not belong together (the plane buffers fail their checksum)")
    model.load_state_dict(blob["state_dict"], strict=True)       # invalidates any earlier pack (post hook)
    model._wide_range = wide
    model._inter_pairs = inter_pairs
    dev = torch.device(device)
    pk = _dec(blob["packed"], dev)
    if tuple(pk["attn_guard"].shape) != (model.cfg["depth"] + 1,):   # (a guard tensor without the clamp word would switch the watch off silently)
        raise RuntimeError(f"{path}: the packed guard words have shape {tuple(pk['attn_guard'].shape)}, expected ({model.cfg['depth'] + 1},): repack")
    # spot check on the device: planes -> float must give back the weights they claim to be the split of
    sd = blob["state_dict"]
    for planes, w in ((pk["pe_w"], sd["patch_embed.proj.weight"].reshape(model.cfg["embed_dim"], -1)),
                      (pk["blocks"][-1]["lin2"], sd[f"blocks.{model.cfg['depth'] - 1}.mlp.lin2.weight"])):
        back = ops.planes_to_float(planes, cols=w.shape[1])[: w.shape[0]].cpu()
        if not torch.allclose(back, w.float(), rtol=2e-3, atol=1e-6):
            raise RuntimeError(f"{path}: state dict and packed planes do not belong together (planes decode to other weights)")
    pk["geom"] = {}
    pk["dev"] = dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
    model._packed = pk
    hd = model.cfg["embed_dim"] // model.cfg["num_heads"]
    model._hd_true, model._hd_pad = hd, ops.pad32(hd)
    return model
