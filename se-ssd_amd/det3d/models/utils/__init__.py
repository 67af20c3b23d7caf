"""mirrors det3d/models/utils: Sequential (misc.py:22-115), build_norm_layer (norm.py:60-111) -- without the
import-time dependency on the syncbn CUDA extension."""
from collections import OrderedDict

import torch
from torch import nn

norm_cfg = {
    "BN": ("bn", nn.BatchNorm2d),
    "BN1d": ("bn1d", nn.BatchNorm1d),
    "GN": ("gn", nn.GroupNorm),
    "SyncBN": ("bn", nn.SyncBatchNorm),  # RCCL-backed torch SyncBatchNorm replaces apex / det3d.ops.syncbn
}


def build_norm_layer(cfg, num_features, postfix=""):
    assert isinstance(cfg, dict) and "type" in cfg
    cfg_ = dict(cfg)
    layer_type = cfg_.pop("type")
    if layer_type not in norm_cfg:
        raise KeyError("Unrecognized norm type {}".format(layer_type))
    abbr, norm_layer = norm_cfg[layer_type]
    name = abbr + str(postfix)
    requires_grad = cfg_.pop("requires_grad", True)
    cfg_.setdefault("eps", 1e-5)
    if layer_type != "GN":
        layer = norm_layer(num_features, **cfg_)
    else:
        layer = norm_layer(num_channels=num_features, **cfg_)
    for param in layer.parameters():
        param.requires_grad = requires_grad
    return name, layer


class Sequential(nn.Module):
    """torch.nn.Sequential with add() and kwargs (misc.py:22-115); integer keys give the reference's state_dict names."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(str(len(self._modules)) if name is None else name, module)

    def forward(self, input):
        for module in self._modules.values():
            input = module(input)
        return input
