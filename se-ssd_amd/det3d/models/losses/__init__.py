"""The three losses config.py:96-111 names for MultiGroupHead (det3d/models/losses/losses.py): elementwise torch
expressions (they run on whatever device their inputs live on) with the reference's constructor kwargs and return shapes.
Pinned by tests/golden/losses_ref.npz = the reference modules run from source. The geometric loss (ODIoU) is the device
op sessd_hip.ops.odiou_3d_loss."""
import torch
import torch.nn.functional as F
from torch import nn

from ..registry import LOSSES


@LOSSES.register_module
class SigmoidFocalLoss(nn.Module):
    """losses.py:365-419. forward(logits (B,A,C), one-hot targets (B,A,C), weights (B,A)) -> (B,A,C):
    alpha_t (1 - p_t)^gamma * BCE-with-logits * weight."""

    def __init__(self, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        self._gamma, self._alpha, self._reduction, self._loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, prediction_tensor, target_tensor, weights=None, class_indices=None):
        w = weights.unsqueeze(2)
        if class_indices is not None:
            sel = torch.zeros(prediction_tensor.shape[2], dtype=prediction_tensor.dtype, device=prediction_tensor.device)
            sel[class_indices] = 1
            w = w * sel.view(1, 1, -1)
        x, t = prediction_tensor, target_tensor.type_as(prediction_tensor)
        bce = torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-torch.abs(x)))
        p = torch.sigmoid(x)
        p_t = t * p + (1 - t) * (1 - p)
        mod = torch.pow(1.0 - p_t, self._gamma) if self._gamma else 1.0
        a_t = t * self._alpha + (1 - t) * (1 - self._alpha) if self._alpha is not None else 1.0
        return mod * a_t * bce * w


@LOSSES.register_module
class WeightedSmoothL1Loss(nn.Module):
    """losses.py:147-203. Smooth L1 with transition at 1/sigma^2: 0.5 (sigma d)^2 below, |d| - 0.5/sigma^2 above; with
    codewise=True the result keeps the code axis: (B,A,code) * weights (B,A,1). The reference ignores code_weights."""

    def __init__(self, sigma=3.0, reduction="mean", code_weights=None, codewise=True, loss_weight=1.0):
        super().__init__()
        self._sigma, self._reduction, self._codewise, self._loss_weight = sigma, reduction, codewise, loss_weight
        self._code_weights = None

    def forward(self, prediction_tensor, target_tensor, weights=None):
        d = torch.abs(prediction_tensor - target_tensor)
        s2 = self._sigma ** 2
        small = (d <= 1.0 / s2).type_as(d)
        loss = small * 0.5 * (d * self._sigma) ** 2 + (d - 0.5 / s2) * (1.0 - small)
        if self._codewise:
            return loss * weights.unsqueeze(-1) if weights is not None else loss
        loss = loss.sum(2)
        return loss * weights if weights is not None else loss


@LOSSES.register_module
class WeightedSoftmaxClassificationLoss(nn.Module):
    """losses.py:498-531. Cross entropy of logits (B,A,C) against the argmax of the one-hot targets, times weights (B,A)."""

    def __init__(self, logit_scale=1.0, loss_weight=1.0, name=""):
        super().__init__()
        self.name, self._loss_weight, self._logit_scale = name, loss_weight, logit_scale

    def forward(self, prediction_tensor, target_tensor, weights):
        c = prediction_tensor.shape[-1]
        logits = (prediction_tensor / self._logit_scale).reshape(-1, c)
        labels = target_tensor.reshape(-1, c).max(dim=-1)[1]
        return F.cross_entropy(logits, labels, reduction="none").view(weights.shape) * weights
