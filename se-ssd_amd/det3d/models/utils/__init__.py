"""Boundary glue of the det3d module API: the container and the norm factory that `rpn_v1.py` / `scn.py` are written against
(reference interface: det3d/models/utils/misc.py:22 `Sequential`, norm.py:60 `build_norm_layer`). Written on top of
torch.nn.Sequential; child names are positional ("0", "1", ...) unless given, which is what makes the state_dict keys of the
mirror equal the reference's (tests/test_det3d_mirror.py holds the key set)."""
import torch.nn as nn


class Sequential(nn.Sequential):
    """nn.Sequential plus the two things the reference's necks use: named children through keyword arguments and `add()`."""

    def __init__(self, *modules, **named):
        super().__init__(*modules)
        for key, mod in named.items():
            self.add_module(key, mod)

    def add(self, module, name=None):
        self.add_module(name if name is not None else str(len(self)), module)


# config `type` -> (attribute prefix, factory(num_features, **kwargs)). SyncBN is torch's own (RCCL all-reduce of the statistics):
# the reference's apex / det3d.ops.syncbn extension has no importer on this path.
_NORMS = {
    "BN": ("bn", nn.BatchNorm2d),
    "BN1d": ("bn1d", nn.BatchNorm1d),
    "SyncBN": ("bn", nn.SyncBatchNorm),
    "GN": ("gn", lambda c, **kw: nn.GroupNorm(num_channels=c, **kw)),
}
norm_cfg = {k: v for k, v in _NORMS.items()}


def build_norm_layer(cfg, num_features, postfix=""):
    """(name, layer) for a config dict like dict(type="BN", eps=1e-3, momentum=0.01); `requires_grad=False` freezes it."""
    kw = dict(cfg)
    kind = kw.pop("type")
    trainable = kw.pop("requires_grad", True)
    kw.setdefault("eps", 1e-5)
    try:
        prefix, make = _NORMS[kind]
    except KeyError:
        raise KeyError("Unrecognized norm type %s" % kind) from None
    layer = make(num_features, **kw)
    layer.requires_grad_(trainable)
    return prefix + str(postfix), layer
