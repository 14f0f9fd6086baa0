"""mmsa -- MI355X-native image-encoder forward of Multimodal-SAM-Adapter behind the reference's mmseg backbone API.

Importing this package loads libmmsa_hip.so (hard requirement; no CPU fallback) and registers the backbone
classes under the reference's names."""
from . import lib  # noqa: F401  (raises if the HIP library is missing)
from . import ops  # noqa: F401
from .backbone import SAMAdapterbimodalMixModNewInTwinConvNEW, SAMAdapterbimodalMixModNewInTwinConvNEWwithcp
from .registry import BACKBONES, build_backbone

for _cls in (SAMAdapterbimodalMixModNewInTwinConvNEW, SAMAdapterbimodalMixModNewInTwinConvNEWwithcp):
    BACKBONES.register_module(force=True)(_cls)

__all__ = ["SAMAdapterbimodalMixModNewInTwinConvNEW", "SAMAdapterbimodalMixModNewInTwinConvNEWwithcp",
           "BACKBONES", "build_backbone", "ops", "lib"]
