"""Registry / build_from_cfg (mirrors det3d/utils/registry.py:6-76)."""
import inspect


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __repr__(self):
        return self.__class__.__name__ + "(name={}, items={})".format(self._name, list(self._module_dict.keys()))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register_module(self, module_class):
        if not inspect.isclass(module_class):
            raise TypeError("module must be a class, but got {}".format(type(module_class)))
        name = module_class.__name__
        if name in self._module_dict:
            raise KeyError("{} is already registered in {}".format(name, self.name))
        self._module_dict[name] = module_class

    def register_module(self, cls):
        self._register_module(cls)
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(cfg, dict) and "type" in cfg
    assert isinstance(default_args, dict) or default_args is None
    args = dict(cfg)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError("{} is not in the {} registry".format(obj_type, registry.name))
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError("type must be a str or valid type, but got {}".format(type(obj_type)))
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    return obj_cls(**args)
