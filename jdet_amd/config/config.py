"""Config loader.  Mirrors python/jdet/config/config.py:L16-165: `.py` / `.yaml` files, `_base_`
inheritance (str or list, relative to the including file), `_cover_` to replace instead of merge,
attribute access returning None for missing keys, `name` / `work_dir` defaults."""
import copy
import inspect
import os
import sys
from collections import OrderedDict
from importlib import import_module

import yaml

__all__ = ["Config", "get_cfg", "init_cfg", "save_cfg", "print_cfg", "update_cfg"]
BASE_KEY = "_base_"
COVER_KEY = "_cover_"


class Config(OrderedDict):
    def __init__(self, *args):
        super().__init__()
        if len(args) == 1:
            self.load_from_file(args[0])
        else:
            assert len(args) == 0

    def __getattr__(self, name):
        if name in self:
            return self[name]
        return None

    def __setattr__(self, name, value):
        self[name] = value

    @staticmethod
    def _load_dict_from_file_no_base(filename):
        assert os.path.isfile(filename), filename
        if filename.endswith(".yaml"):
            with open(filename, "r") as f:
                cfg = yaml.safe_load(f.read())
        elif filename.endswith(".py"):
            f_dir = os.path.dirname(os.path.abspath(filename))
            module_name = os.path.basename(filename)[:-3]
            sys.path.insert(0, f_dir)
            try:
                sys.modules.pop(module_name, None)
                mod = import_module(module_name)
            finally:
                sys.path.pop(0)
            cfg = {name: value for name, value in mod.__dict__.items() if not name.startswith("__")}
            del sys.modules[module_name]
        else:
            assert False, "unsupported config type."
        return cfg

    @staticmethod
    def _load_dict_from_file(filename):
        cfg = Config._load_dict_from_file_no_base(filename)
        cfg_dir = os.path.dirname(filename)
        if BASE_KEY in cfg:
            if isinstance(cfg[BASE_KEY], list):
                base_filenames = cfg[BASE_KEY]
            else:
                assert isinstance(cfg[BASE_KEY], str)
                base_filenames = [cfg[BASE_KEY]]
            cfg_base = {}
            for bfn in base_filenames:
                Config.merge_dict_b2a(cfg_base, Config._load_dict_from_file(os.path.join(cfg_dir, bfn)))
            cfg.pop(BASE_KEY)
            Config.merge_dict_b2a(cfg_base, cfg)
            cfg = cfg_base
        return cfg

    @staticmethod
    def merge_dict_b2a(a, b):
        def clear_cover_key(x):
            if not isinstance(x, dict):
                return x
            out = copy.deepcopy(x)
            if COVER_KEY in out:
                out.pop(COVER_KEY)
            for k, v in out.items():
                out[k] = clear_cover_key(v)
            return out

        assert isinstance(a, dict) and isinstance(b, dict)
        if COVER_KEY in b:
            a.clear()
            a.update(clear_cover_key(copy.deepcopy(b)))
            return
        for k, v in b.items():
            if (k not in a) or (isinstance(v, dict) and v.get(COVER_KEY, False)) or (not isinstance(v, dict)) or (
                    not isinstance(a[k], dict)):
                a[k] = clear_cover_key(copy.deepcopy(v))
            else:
                Config.merge_dict_b2a(a[k], v)

    def load_from_file(self, filename):
        cfg = Config._load_dict_from_file(filename)
        self.clear()
        self.update(self.dfs(cfg))
        if self.name is None:
            self.name = os.path.splitext(os.path.basename(filename))[0]
        if self.work_dir is None:
            self.work_dir = f"work_dirs/{self.name}"

    def dfs(self, cfg_other):
        if isinstance(cfg_other, dict):
            now_cfg = Config()
            for k, d in cfg_other.items():
                if inspect.ismodule(d):
                    continue
                now_cfg[k] = self.dfs(d)
        elif isinstance(cfg_other, list):
            now_cfg = [self.dfs(d) for d in cfg_other if not inspect.ismodule(d)]
        else:
            now_cfg = copy.deepcopy(cfg_other)
        return now_cfg

    def dump(self):
        now = dict()
        for k, d in self.items():
            if isinstance(d, Config):
                d = d.dump()
            if isinstance(d, list):
                d = [dd.dump() if isinstance(dd, Config) else dd for dd in d]
            now[k] = d
        return now


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
