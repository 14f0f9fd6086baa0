"""mmseg `BACKBONES` registration (BK:27-28, backbones/__init__.py:3-9).  When mmseg is importable the classes
are registered into its registry, so `build_backbone(cfg)` (ED:36) resolves `type='SAMAdapterbimodalMixModNewInTwinConvNEW'`
to the MI355X implementation; otherwise a minimal local registry with the same `register_module` / `build`
surface is provided."""


class _LocalRegistry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg):
        cfg = dict(cfg)
        t = cfg.pop("type")
        cls = self.get(t)
        if cls is None:
            raise KeyError(f"{t} is not in the {self.name} registry")
        return cls(**cfg)


try:  # pragma: no cover - mmseg is not installed in the build image
    from mmseg.models.builder import BACKBONES, HEADS  # type: ignore
    HAVE_MMSEG = True
except Exception:  # noqa: BLE001
    BACKBONES = _LocalRegistry("backbone")
    HEADS = _LocalRegistry("head")
    HAVE_MMSEG = False


def build_backbone(cfg):
    """Equivalent of mmseg.models.builder.build_backbone for the local registry."""
    return BACKBONES.build(cfg)


def build_head(cfg):
    """Equivalent of mmseg.models.builder.build_head (ED:44) for the local registry."""
    return HEADS.build(cfg)
