"""mmsa -- MI355X-native image-encoder forward of Multimodal-SAM-Adapter behind the reference's mmseg backbone API.

Importing this package loads libmmsa_hip.so (hard requirement; no CPU fallback) and registers the backbone
classes under the reference's names."""
from . import lib  # noqa: F401  (raises if the HIP library is missing)
from . import ops  # noqa: F401
from . import inference  # noqa: F401  (encode_decode / slide_inference / argmax map of the reference's EncoderDecoder)
from .backbone import SAMAdapterbimodalMixModNewInTwinConvNEW, SAMAdapterbimodalMixModNewInTwinConvNEWwithcp
from .head import SegformerHead
from .registry import BACKBONES, HEADS, build_backbone, build_head

for _cls in (SAMAdapterbimodalMixModNewInTwinConvNEW, SAMAdapterbimodalMixModNewInTwinConvNEWwithcp):
    BACKBONES.register_module(force=True)(_cls)

HEADS.register_module(force=True)(SegformerHead)  # segformer_head.py:11 registers with force=True as well

__all__ = ["SegformerHead", "HEADS", "build_head", "SAMAdapterbimodalMixModNewInTwinConvNEW", "SAMAdapterbimodalMixModNewInTwinConvNEWwithcp",
           "BACKBONES", "build_backbone", "ops", "lib", "inference"]
