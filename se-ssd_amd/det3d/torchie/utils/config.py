"""Config / ConfigDict with attribute access (mirrors det3d/torchie/utils/config.py:12-29,59-160), without addict.

Non-dict values (the box-coder object, a logging.Logger) survive untouched inside the tree; missing keys raise
KeyError / AttributeError exactly like the reference's ConfigDict.__missing__."""
import os.path as osp
import sys
from importlib import import_module


class ConfigDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _hook(cls, item):
        if isinstance(item, dict) and not isinstance(item, ConfigDict):
            return cls(item)
        if isinstance(item, (list, tuple)):
            return type(item)(cls._hook(e) for e in item)
        return item

    def __setitem__(self, name, value):
        super().__setitem__(name, self._hook(value))

    def __missing__(self, name):
        raise KeyError(name)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'{}' object has no attribute '{}'".format(self.__class__.__name__, name))

    def __setattr__(self, name, value):
        self[name] = value

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}


class Config(object):
    @staticmethod
    def fromfile(filename):
        filename = osp.abspath(osp.expanduser(filename))
        if not osp.isfile(filename):
            raise FileNotFoundError('file "{}" does not exist'.format(filename))
        if not filename.endswith(".py"):
            raise IOError("Only py type is supported by this mirror")
        module_name = osp.basename(filename)[:-3]
        if "." in module_name:
            raise ValueError("Dots are not allowed in config file path.")
        config_dir = osp.dirname(filename)
        sys.path.insert(0, config_dir)
        try:
            sys.modules.pop(module_name, None)
            mod = import_module(module_name)
        finally:
            sys.path.pop(0)
        cfg_dict = {name: value for name, value in mod.__dict__.items() if not name.startswith("__")}
        return Config(cfg_dict, filename=filename)

    def __init__(self, cfg_dict=None, filename=None):
        if cfg_dict is None:
            cfg_dict = dict()
        elif not isinstance(cfg_dict, dict):
            raise TypeError("cfg_dict must be a dict, but got {}".format(type(cfg_dict)))
        super().__setattr__("_cfg_dict", ConfigDict(cfg_dict))
        super().__setattr__("_filename", filename)
        text = ""
        if filename:
            with open(filename, "r") as f:
                text = f.read()
        super().__setattr__("_text", text)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    def __repr__(self):
        return "Config (path: {}): {}".format(self.filename, self._cfg_dict.__repr__())

    def __len__(self):
        return len(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict.__getitem__(name)

    def __setattr__(self, name, value):
        self._cfg_dict.__setattr__(name, value)

    def __setitem__(self, name, value):
        self._cfg_dict.__setitem__(name, value)

    def __iter__(self):
        return iter(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)
