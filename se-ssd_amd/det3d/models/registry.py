"""The six model registries of det3d/models/registry.py, created from one table."""
from det3d.utils import Registry

_KINDS = dict(READERS="reader", BACKBONES="backbone", NECKS="neck", HEADS="head", LOSSES="loss", DETECTORS="detector")
globals().update({var: Registry(label) for var, label in _KINDS.items()})
__all__ = sorted(_KINDS)
