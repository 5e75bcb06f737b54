"""Config files of the reference, loaded unchanged.

Contract of python/jdet/config/config.py:L16-165: `.py` and `.yaml` files; `_base_` (a path or a list of paths,
relative to the including file) is loaded first and the including file merged over it, dict by dict; a dict carrying
`_cover_: True` replaces instead of merging; nested dicts become `Config` objects whose missing attributes read as
None; `name` defaults to the file stem and `work_dir` to `work_dirs/<name>`; one process-wide instance behind
`init_cfg` / `get_cfg` / `update_cfg` / `save_cfg` / `print_cfg`.
"""
import copy
import os
import runpy
import types
from collections import OrderedDict

import yaml

__all__ = ["Config", "get_cfg", "init_cfg", "save_cfg", "print_cfg", "update_cfg"]
BASE_KEY = "_base_"
COVER_KEY = "_cover_"


def _read_file(path):
    """one file -> plain dict (python configs: the module namespace without dunders and imported modules)"""
    assert os.path.isfile(path), path
    ext = os.path.splitext(path)[1]
    if ext == ".yaml":
        with open(path, "r") as f:
            return yaml.safe_load(f.read())
    assert ext == ".py", "unsupported config type."
    ns = runpy.run_path(os.path.abspath(path))
    return {k: v for k, v in ns.items() if not k.startswith("__") and not isinstance(v, types.ModuleType)}


def _without_cover_marks(value):
    if not isinstance(value, dict):
        return copy.deepcopy(value)
    return {k: _without_cover_marks(v) for k, v in value.items() if k != COVER_KEY}


def _merge(dst, src):
    """src over dst, in place.  A dict marked `_cover_` replaces; dicts merge recursively; anything else overwrites."""
    assert isinstance(dst, dict) and isinstance(src, dict)
    if COVER_KEY in src:
        dst.clear()
        dst.update(_without_cover_marks(src))
        return dst
    for key, value in src.items():
        both_dicts = isinstance(value, dict) and isinstance(dst.get(key), dict)
        if both_dicts and not value.get(COVER_KEY, False):
            _merge(dst[key], value)
        else:
            dst[key] = _without_cover_marks(value)
    return dst


def _load_with_bases(path):
    own = _read_file(path)
    bases = own.pop(BASE_KEY, None)
    if bases is None:
        return own
    if isinstance(bases, str):
        bases = [bases]
    assert isinstance(bases, list)
    merged = {}
    for rel in bases:
        _merge(merged, _load_with_bases(os.path.join(os.path.dirname(path), rel)))
    return _merge(merged, own)


class Config(OrderedDict):
    def __init__(self, *args):
        super().__init__()
        assert len(args) <= 1
        if args:
            self.load_from_file(args[0])

    def __getattr__(self, name):
        return self[name] if name in self else None

    def __setattr__(self, name, value):
        self[name] = value

    @classmethod
    def _wrap(cls, value):
        if isinstance(value, dict):
            node = cls()
            for k, v in value.items():
                node[k] = cls._wrap(v)
            return node
        if isinstance(value, list):
            return [cls._wrap(v) for v in value]
        return copy.deepcopy(value)

    def load_from_file(self, filename):
        tree = self._wrap(_load_with_bases(filename))
        self.clear()
        self.update(tree)
        if self.name is None:
            self.name = os.path.splitext(os.path.basename(filename))[0]
        if self.work_dir is None:
            self.work_dir = f"work_dirs/{self.name}"

    def dump(self):
        def plain(v):
            if isinstance(v, Config):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, list):
                return [plain(x) for x in v]
            return v
        return plain(self)


_cfg = Config()


def init_cfg(filename):
    print("Loading config from: ", filename)
    _cfg.load_from_file(filename)


def get_cfg():
    return _cfg


def update_cfg(args):
    _cfg.update(args)


def save_cfg(save_file):
    with open(save_file, "w") as f:
        f.write(yaml.dump(_cfg.dump()))


def print_cfg():
    print(yaml.dump(_cfg.dump()))
