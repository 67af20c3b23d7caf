"""build_reader / build_backbone / build_neck / build_head / build_loss / build_detector (det3d/models/builder.py:120-150):
a config dict (or a list of them -> nn.Sequential) is turned into the registered module."""
from torch import nn

from det3d.utils import build_from_cfg

from . import registry as _reg


def build(cfg, registry, default_args=None):
    if isinstance(cfg, (list, tuple)):
        return nn.Sequential(*(build_from_cfg(c, registry, default_args) for c in cfg))
    return build_from_cfg(cfg, registry, default_args)


def _builder(registry):
    def _build(cfg):
        return build(cfg, registry)
    _build.__doc__ = "build a %s from its config dict" % registry.name
    return _build


build_reader = _builder(_reg.READERS)
build_backbone = _builder(_reg.BACKBONES)
build_neck = _builder(_reg.NECKS)
build_head = _builder(_reg.HEADS)
build_loss = _builder(_reg.LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, _reg.DETECTORS, {"train_cfg": train_cfg, "test_cfg": test_cfg})
