"""Config / ConfigDict with attribute access (mirrors det3d/torchie/utils/config.py:12-29,59-160), without addict.

Non-dict values (the box-coder object, a logging.Logger) survive untouched inside the tree; missing keys raise
KeyError / AttributeError exactly like the reference's ConfigDict.__missing__."""
import os.path as osp
import sys


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


class Config:
    """The reference's config object as far as this path uses it (reference surface: det3d/torchie/utils/config.py:59-160 --
    `Config.fromfile(path)`, attribute and item access into the file's top-level names, `.filename`, `.text`, `len`, iteration,
    `.get`). A config file is plain Python: it is executed with runpy and its public names become the tree."""

    _OWN = ("_cfg_dict", "_filename", "_text")

    def __init__(self, cfg_dict=None, filename=None):
        if cfg_dict is not None and not isinstance(cfg_dict, dict):
            raise TypeError("cfg_dict must be a dict, but got {}".format(type(cfg_dict)))
        text = ""
        if filename:
            with open(filename) as fh:
                text = fh.read()
        for name, value in zip(self._OWN, (ConfigDict(cfg_dict or {}), filename, text)):
            object.__setattr__(self, name, value)

    @staticmethod
    def fromfile(filename):
        import runpy
        path = osp.abspath(osp.expanduser(filename))
        if not osp.isfile(path):
            raise FileNotFoundError('file "{}" does not exist'.format(path))
        if not path.endswith(".py"):
            raise IOError("Only py type is supported by this mirror")
        if "." in osp.basename(path)[:-3]:
            raise ValueError("Dots are not allowed in config file path.")
        sys.path.insert(0, osp.dirname(path))   # a config may import its neighbours
        try:
            names = runpy.run_path(path)
        finally:
            sys.path.pop(0)
        return Config({k: v for k, v in names.items() if not k.startswith("__")}, filename=path)

    filename = property(lambda self: self._filename)
    text = property(lambda self: self._text)

    def __getattr__(self, name):          # (only reached for names that are not the object's own)
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        setattr(self._cfg_dict, name, value)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __repr__(self):
        return "Config (path: {}): {!r}".format(self._filename, self._cfg_dict)
