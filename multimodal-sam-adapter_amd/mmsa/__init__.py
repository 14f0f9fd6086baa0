"""mmsa -- MI355X-native image-encoder forward of Multimodal-SAM-Adapter behind the reference's mmseg backbone API.

Importing this package loads libmmsa_hip.so (hard requirement; no CPU fallback) and registers the backbone
classes under the reference's names."""
from . import lib  # noqa: F401  (raises if the HIP library is missing)
from . import ops  # noqa: F401
from . import inference  # noqa: F401  (encode_decode / slide_inference / argmax map of the reference's EncoderDecoder)
from .backbone import OperandRangeError, SAMAdapterbimodalMixModNewInTwinConvNEW, SAMAdapterbimodalMixModNewInTwinConvNEWwithcp
from .head import SegformerHead
from .chains import AttentionRangeError, Chains, Replay
from .registry import BACKBONES, HEADS, build_backbone, build_head

for _cls in (SAMAdapterbimodalMixModNewInTwinConvNEW, SAMAdapterbimodalMixModNewInTwinConvNEWwithcp):
    BACKBONES.register_module(force=True)(_cls)



def register_head():
    """Register this package's SegformerHead under the reference's name in HEADS (segformer_head.py:11 registers with force=True
    as well).  With mmseg installed this is OPT-IN: the backbone alone is the drop-in of BASELINE's path and the reference's own
    head keeps working behind it; an import-time override would depend on whether mmseg_custom is imported before or after mmsa."""
    HEADS.register_module(force=True)(SegformerHead)
    return SegformerHead


from .registry import HAVE_MMSEG as _HAVE_MMSEG  # noqa: E402

if not _HAVE_MMSEG:     # local registry (no mmseg in the process): nothing to override, mmsa.build_head works out of the box
    register_head()

__all__ = ["SegformerHead", "HEADS", "build_head", "register_head", "SAMAdapterbimodalMixModNewInTwinConvNEW", "SAMAdapterbimodalMixModNewInTwinConvNEWwithcp",
           "BACKBONES", "build_backbone", "ops", "lib", "inference", "Chains", "Replay", "AttentionRangeError", "OperandRangeError"]
