"""Name -> class registries and config-driven construction (same contract as det3d/utils/registry.py:6-76:
`Registry(name)`, `.register_module` as a class decorator, `.get`, `.module_dict`, `build_from_cfg(cfg, registry,
default_args)` where cfg["type"] is a registered name or a class and the remaining keys become kwargs)."""
import inspect


class Registry(object):
    def __init__(self, name):
        self._name = str(name)
        self._classes = {}

    # -- read access -------------------------------------------------------------------------------------
    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._classes)

    def get(self, key):
        return self._classes.get(key)

    def __contains__(self, key):
        return key in self._classes

    def __len__(self):
        return len(self._classes)

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (type(self).__name__, self._name, sorted(self._classes))

    # -- registration -------------------------------------------------------------------------------------
    def register_module(self, cls):
        """Class decorator. Registering two classes under one name is an error (KeyError), like the reference."""
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got %s" % type(cls))
        key = cls.__name__
        if key in self._classes:
            raise KeyError("%s is already registered in %s" % (key, self._name))
        self._classes[key] = cls
        return cls


def _resolve(kind, registry):
    if inspect.isclass(kind):
        return kind
    if not isinstance(kind, str):
        raise TypeError("type must be a str or valid type, but got %s" % type(kind))
    cls = registry.get(kind)
    if cls is None:
        raise KeyError("%s is not in the %s registry" % (kind, registry.name))
    return cls


def build_from_cfg(cfg, registry, default_args=None):
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise AssertionError("cfg must be a dict with a 'type' key")
    if default_args is not None and not isinstance(default_args, dict):
        raise AssertionError("default_args must be a dict or None")
    kwargs = {k: v for k, v in cfg.items() if k != "type"}
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return _resolve(cfg["type"], registry)(**kwargs)
