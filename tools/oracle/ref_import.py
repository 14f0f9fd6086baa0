"""Import harness for the *reference* backbone (runs ONLY where /root/reference exists).

This is the build's own stub code: it installs permissive stand-ins for the
third-party packages the reference imports but this image lacks (mmcv, mmseg,
mmdet, timm, cv2, ...), then imports the reference's registered backbone class
`SAMAdapterbimodalMixModNewInTwinConvNEW` from
`segmentation/mmseg_custom/models/backbones/image_encoder_adapter_bimodal_mix_mod_new_in_twin_convnext_new.py:27-28`
UNMODIFIED, with the native MSDA op replaced by the reference's own
`ms_deform_attn_core_pytorch` (`segmentation/ops/functions/ms_deform_attn_func.py:53-75`),
which the reference's `ops/test.py:26-75` declares equivalent to its CUDA kernel.

No reference source or bytecode is copied anywhere; golden *vectors* produced through
this harness are committed under tests/golden/ by tools/oracle/make_golden.py.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types
from unittest import mock

REF_ROOT = "/root/reference"
SEG = os.path.join(REF_ROOT, "segmentation")

_MOCK_ROOTS = ("mmdet", "mmcv", "mmseg", "mmcls", "timm", "cv2", "torchvision",
               "yapf", "termcolor", "pavi", "matplotlib", "tensorboard")


def available() -> bool:
    return os.path.isdir(SEG)


class _Registry:
    def __init__(self, name="reg"):
        self.name = name
        self.scope = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name if isinstance(name, str) else cls.__name__
            self.module_dict[key] = cls
            return cls
        if module is not None:
            return deco(module)
        if isinstance(name, type):  # used as bare decorator
            cls, name = name, None
            return deco(cls)
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, *a, **k):
        cfg = dict(cfg)
        t = cfg.pop("type")
        return self.module_dict[t](**cfg)


class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


class _MockFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _MOCK_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, _MockLoader(), is_package=True)
        return None


def _mk(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install(withcp=False):
    """Install stubs and return the reference backbone class (`withcp`: ...NEWwithcp, image_encoder_adapter_..._new_with_cp.py:28)."""
    global _installed
    import torch
    import torch.nn as nn
    if not available():
        raise RuntimeError("reference tree not present; goldens can only be generated in the build container")
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if not _installed:
        sys.meta_path.insert(0, _MockFinder())
        os.chdir(SEG)
        sys.path.insert(0, SEG)

        # addict.Dict (real recursive attribute dict)
        class Dict(dict):
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    raise AttributeError(k)

            def __setattr__(self, k, v):
                self[k] = v
        _mk("addict", Dict=Dict)

        # registries
        import mmseg.models.builder as b  # mock module
        b.BACKBONES = _Registry("backbone")
        b.HEADS = _Registry("head")
        b.SEGMENTORS = _Registry("segmentor")
        b.LOSSES = _Registry("loss")

        # timm.models.layers: three real symbols
        class DropPath(nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                if self.drop_prob == 0.0 or not self.training:
                    return x
                keep = 1 - self.drop_prob
                shape = (x.shape[0],) + (1,) * (x.ndim - 1)
                r = x.new_empty(shape).bernoulli_(keep)
                return x * r / keep

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        import timm.models.layers as tl
        tl.DropPath = DropPath
        tl.trunc_normal_ = nn.init.trunc_normal_
        tl.to_2tuple = to_2tuple

        import mmcv.utils as mu
        mu.TORCH_VERSION = torch.__version__
        mu._BatchNorm = torch.nn.modules.batchnorm._BatchNorm
        import mmcv.runner as mr
        mr.HOOKS = _Registry("hooks")
        mr.OPTIMIZER_BUILDERS = _Registry("ob")
        mr.DefaultOptimizerConstructor = object
        mr.EpochBasedRunner = object
        mr.TextLoggerHook = object
        import mmcv.runner.builder as mrb
        mrb.RUNNERS = _Registry("runners")
        import mmcv.runner.hooks as mrh
        mrh.HOOKS = mr.HOOKS
        mrh.Hook = object
        mrh.OptimizerHook = object

        # bypass heavyweight package __init__s of mmseg_custom
        for pkg, rel in (("mmseg_custom", "mmseg_custom"),
                         ("mmseg_custom.models", "mmseg_custom/models"),
                         ("mmseg_custom.models.backbones", "mmseg_custom/models/backbones"),
                         ("mmseg_custom.models.backbones.base", "mmseg_custom/models/backbones/base")):
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(SEG, rel)]
            sys.modules[pkg] = m
        sys.modules["MultiScaleDeformableAttention"] = types.ModuleType("MultiScaleDeformableAttention")
        _installed = True

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        mod = importlib.import_module(
            "mmseg_custom.models.backbones.image_encoder_adapter_bimodal_mix_mod_new_in_twin_convnext_new")
    import ops.modules.ms_deform_attn as msda_mod
    from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch

    class _Shim:
        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            return ms_deform_attn_core_pytorch(value, shapes, loc, w)
    msda_mod.MSDeformAttnFunction = _Shim
    if withcp:   # the second registered name (backbones/__init__.py:3-9): the FMB configs' class, its own module and adapter-modules file
        with contextlib.redirect_stdout(io.StringIO()):
            mod = importlib.import_module(
                "mmseg_custom.models.backbones.image_encoder_adapter_bimodal_mix_mod_new_in_twin_convnext_new_with_cp")
        return mod.SAMAdapterbimodalMixModNewInTwinConvNEWwithcp
    return mod.SAMAdapterbimodalMixModNewInTwinConvNEW


def ref_functions():
    """Reference free functions used for per-op goldens."""
    install()
    from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    ie = importlib.import_module("mmseg_custom.models.backbones.base.image_encoder")
    am = importlib.import_module(
        "mmseg_custom.models.backbones.adapter_modules_multimodal_mix_mod_new_in_twin_convnext_new")
    return dict(msda_core=ms_deform_attn_core_pytorch,
                window_partition=ie.window_partition, window_unpartition=ie.window_unpartition,
                get_rel_pos=ie.get_rel_pos, add_decomposed_rel_pos=ie.add_decomposed_rel_pos,
                deform_inputs=am.deform_inputs, image_encoder=ie, adapter_modules=am)


def build_reference(**cfg):
    """Construct the reference backbone (eval mode). `checkpoint` is pointed at a dummy
    local file because TwinConvNeXt.init_weights (base/twin_convnext.py:403-443) insists
    on loading one."""
    import contextlib
    import io
    import torch
    cfg = dict(cfg)
    cls = install(withcp=bool(cfg.pop("_withcp", False)))
    tmp = os.path.join(tempfile.gettempdir(), "mmsa_dummy_convnext.pth")
    if not os.path.exists(tmp):
        torch.save({"state_dict": {"dummy.key": torch.zeros(1)}}, tmp)
    cfg = dict(cfg)
    cfg.setdefault("checkpoint", tmp)
    cfg.setdefault("pretrained", None)
    with contextlib.redirect_stdout(io.StringIO()):
        model = cls(**cfg)
    model.eval()
    return model


def build_reference_head(**cfg):
    """Construct the reference's `SegformerHead` (decode_heads/segformer_head.py:11-66, imported UNMODIFIED).

    Its base class `mmseg.models.decode_heads.decode_head.BaseDecodeHead` (mmsegmentation==0.20.2, README.md:98-111)
    is not vendored and not installed, so the harness supplies the ~30 lines of it that the inference path touches
    (constructor attributes, `_transform_inputs('multiple_select')`, `cls_seg` = Dropout2d + 1x1 `conv_seg`), restated
    from the published 0.20.2 behaviour.  `mmcv.cnn.ConvModule` is served by the reference's own vendored
    `mmcv_custom.cnn.ConvModule`; `mmseg.ops.resize` is the `F.interpolate` wrapper it is upstream."""
    import contextlib
    import io
    import torch.nn as nn
    import torch.nn.functional as F
    install()

    class BaseDecodeHead(nn.Module):
        def __init__(self, in_channels, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                     act_cfg=dict(type="ReLU"), in_index=-1, input_transform=None, loss_decode=None, ignore_index=255,
                     sampler=None, align_corners=False, init_cfg=None):
            super().__init__()
            assert input_transform == "multiple_select" and len(in_channels) == len(in_index)
            self.input_transform, self.in_index, self.in_channels = input_transform, in_index, in_channels
            self.channels, self.num_classes, self.dropout_ratio = channels, num_classes, dropout_ratio
            self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
            self.ignore_index, self.align_corners = ignore_index, align_corners
            self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
            self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None

        def _transform_inputs(self, inputs):
            return [inputs[i] for i in self.in_index]

        def cls_seg(self, feat):
            if self.dropout is not None:
                feat = self.dropout(feat)
            return self.conv_seg(feat)

    def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True):
        return F.interpolate(input, size, scale_factor, mode, align_corners)

    import mmseg.models.decode_heads.decode_head as dh  # mock module
    dh.BaseDecodeHead = BaseDecodeHead
    import mmseg.ops as mo
    mo.resize = resize
    from mmcv_custom.cnn import ConvModule  # the reference's vendored copy
    import mmcv.cnn as mc
    mc.ConvModule = ConvModule
    if "mmseg_custom.models.decode_heads" not in sys.modules:
        m = types.ModuleType("mmseg_custom.models.decode_heads")
        m.__path__ = [os.path.join(SEG, "mmseg_custom/models/decode_heads")]
        sys.modules["mmseg_custom.models.decode_heads"] = m
    with contextlib.redirect_stdout(io.StringIO()):
        mod = importlib.import_module("mmseg_custom.models.decode_heads.segformer_head")
        head = mod.SegformerHead(**cfg)
    head.eval()
    return head


def reference_slide_inference(encode_decode_fn, img, crop_size, stride, num_classes):
    """Run the reference's own `EncoderDecoder.slide_inference` (segmentors/encoder_decoder.py:191-234, UNMODIFIED) with a stand-in
    `self` carrying test_cfg / num_classes / align_corners and the given encode_decode callable; rescale=False."""
    import torch.nn as nn
    install()
    if "mmseg_custom.models.segmentors" not in sys.modules:
        m = types.ModuleType("mmseg_custom.models.segmentors")
        m.__path__ = [os.path.join(SEG, "mmseg_custom/models/segmentors")]
        sys.modules["mmseg_custom.models.segmentors"] = m
    import mmseg.models.segmentors.base as sb

    class BaseSegmentor(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
    sb.BaseSegmentor = BaseSegmentor
    mod = importlib.import_module("mmseg_custom.models.segmentors.encoder_decoder")
    fake = types.SimpleNamespace(test_cfg=types.SimpleNamespace(stride=stride, crop_size=crop_size), num_classes=num_classes,
                                 align_corners=False, encode_decode=lambda im, meta: encode_decode_fn(im))
    return mod.EncoderDecoder.slide_inference(fake, img, [dict(ori_shape=tuple(img.shape[2:]) + (3,))], False)


def reference_whole_dim(encode_decode_fn, img, dim, cut_dim=None, rescale=True):
    """Run the reference's own `EncoderDecoder.whole_inference_dim` (segmentors/encoder_decoder.py:329-362) or, with `cut_dim`,
    `whole_inference_dim_cut` (:364-413), UNMODIFIED, with a stand-in `self` whose `encode_decode_test` returns what the reference's
    does behind this backbone: the logits at input size plus the backbone's second return value `(None,)` (:96-107, BK:349)."""
    import torch.nn as nn
    install()
    if "mmseg_custom.models.segmentors" not in sys.modules:
        m = types.ModuleType("mmseg_custom.models.segmentors")
        m.__path__ = [os.path.join(SEG, "mmseg_custom/models/segmentors")]
        sys.modules["mmseg_custom.models.segmentors"] = m
    import mmseg.models.segmentors.base as sb
    import mmseg.ops as mo
    import torch.nn.functional as F

    class BaseSegmentor(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
    sb.BaseSegmentor = BaseSegmentor
    mo.resize = lambda input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True: F.interpolate(input, size, scale_factor, mode, align_corners)
    mod = importlib.import_module("mmseg_custom.models.segmentors.encoder_decoder")
    mod.resize = mo.resize
    fake = types.SimpleNamespace(align_corners=False, encode_decode_test=lambda im, meta: (encode_decode_fn(im), (None,)))
    meta = [dict(ori_shape=tuple(img.shape[2:]) + (3,))]
    if cut_dim is None:
        return mod.EncoderDecoder.whole_inference_dim(fake, img, meta, rescale, dim)
    return mod.EncoderDecoder.whole_inference_dim_cut(fake, img, meta, rescale, dim, cut_dim)
