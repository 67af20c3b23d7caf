"""Loss registry entries named by config.py. Inference never calls them; they carry their kwargs so that
MultiGroupHead builds exactly as in the reference (training-step kernels are SURVEY section 8f 'next' rows)."""
from torch import nn

from ..registry import LOSSES


class _ConfiguredLoss(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.cfg = dict(kwargs)
        for k, v in kwargs.items():
            setattr(self, k if not hasattr(self, k) else "_" + k, v)

    def forward(self, *a, **k):
        raise NotImplementedError("%s: the training step is outside the inference hot path of this build" % type(self).__name__)


@LOSSES.register_module
class SigmoidFocalLoss(_ConfiguredLoss):
    pass


@LOSSES.register_module
class WeightedSmoothL1Loss(_ConfiguredLoss):
    pass


@LOSSES.register_module
class WeightedSoftmaxClassificationLoss(_ConfiguredLoss):
    pass
