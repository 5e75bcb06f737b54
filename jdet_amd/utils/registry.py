"""Name -> class tables and the config-driven constructor.

Same contract as python/jdet/utils/registry.py:L1-63: the fourteen registry objects by the same names,
`@REG.register_module()` / `@REG.register_module(name)` / `REG.register_module(module=cls)`, and
`build_from_cfg(cfg, registry, **kw)` accepting a type string, a dict with a `type` key (remaining keys become
constructor arguments, `kw` wins), a list (-> nn.Sequential of the built items) or None (-> None).  Duplicate and
unknown names fail with AssertionError, a bad constructor call with a TypeError naming the class, as there.
"""
from torch import nn

_REGISTRY_NAMES = ("DATASETS", "TRANSFORMS", "MODELS", "BACKBONES", "HEADS", "LOSSES", "OPTIMS", "BRICKS", "NECKS",
                   "SCHEDULERS", "BOXES", "HOOKS", "ROI_EXTRACTORS", "SHARED_HEADS")


class Registry:
    """a dict of constructors with decorator-style registration"""

    def __init__(self, label=""):
        self.label = label
        self._modules = {}

    def _add(self, obj, key):
        key = obj.__name__ if key is None else key
        assert key not in self._modules, f"{key} is already registered."
        self._modules[key] = obj
        return obj

    def register_module(self, name=None, module=None):
        if module is None:                      # used as a decorator (with or without an explicit name)
            return lambda obj: self._add(obj, name)
        return self._add(module, name)

    def get(self, name):
        assert name in self._modules, f"{name} is not registered."
        return self._modules[name]

    def __contains__(self, name):
        return name in self._modules

    def __len__(self):
        return len(self._modules)

    def __repr__(self):
        return "Registry(%s: %d entries)" % (self.label, len(self._modules))


def _construct(cls, kwargs):
    try:
        return cls(**kwargs)
    except TypeError as err:                    # say which class rejected the arguments
        msg = str(err)
        raise TypeError(msg if "<class" in msg else f"{cls}.{msg}")


def build_from_cfg(cfg, registry, **kwargs):
    if cfg is None:
        return None
    if isinstance(cfg, list):
        return nn.Sequential(*(build_from_cfg(item, registry, **kwargs) for item in cfg))
    if isinstance(cfg, str):
        return registry.get(cfg)(**kwargs)
    if isinstance(cfg, dict):
        params = {k: v for k, v in cfg.items() if k != "type"}
        params.update(kwargs)
        return _construct(registry.get(cfg["type"]), params)
    raise TypeError(f"type {type(cfg)} not support")


globals().update({_n: Registry(_n) for _n in _REGISTRY_NAMES})
__all__ = ["Registry", "build_from_cfg", *_REGISTRY_NAMES]
