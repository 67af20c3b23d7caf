"""Generates tests/golden/losses_ref.npz by running the REFERENCE's loss modules from source on CPU:
    det3d/models/losses/losses.py  SigmoidFocalLoss :365-419, WeightedSmoothL1Loss :147-203,
                                   WeightedSoftmaxClassificationLoss :498-531
(the three losses config.py:96-111 names for MultiGroupHead). losses.py uses package-relative imports, so it is loaded
under a stub package whose registry decorator is the identity.
Run in the build container only (needs /root/reference):  python tests/golden/make_golden_losses.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_losses():
    for name in ("refpkg", "refpkg.models", "refpkg.models.losses"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    reg = types.ModuleType("refpkg.models.registry")

    class _Reg:
        @staticmethod
        def register_module(obj):
            return obj
    reg.LOSSES = _Reg()
    sys.modules["refpkg.models.registry"] = reg
    for sub in ("utils", "losses"):
        spec = importlib.util.spec_from_file_location("refpkg.models.losses." + sub,
                                                      os.path.join(REF, "det3d/models/losses", sub + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["refpkg.models.losses." + sub] = mod
        spec.loader.exec_module(mod)
    return sys.modules["refpkg.models.losses.losses"]


def main():
    assert os.path.isdir(REF)
    L = load_losses()
    g = torch.Generator().manual_seed(0)
    B, A = 2, 600
    out = {}
    # focal: logits (B,A,1), one-hot targets (B,A,1), weights (B,A)
    logits = torch.randn(B, A, 1, generator=g) * 3
    targets = (torch.rand(B, A, 1, generator=g) < 0.05).float()
    w = torch.rand(B, A, generator=g)
    lr = logits.clone().requires_grad_(True)
    y = L.SigmoidFocalLoss(gamma=2.0, alpha=0.25, loss_weight=1.0)(lr, targets, weights=w.clone())
    y.sum().backward()
    out.update(focal_logits=logits.numpy(), focal_targets=targets.numpy(), focal_w=w.numpy(), focal_out=y.detach().numpy(),
               focal_grad=lr.grad.numpy())
    # smooth L1 (sigma 3, codewise): (B,A,7)
    pred = torch.randn(B, A, 7, generator=g) * 0.3
    tgt = torch.randn(B, A, 7, generator=g) * 0.3
    pr = pred.clone().requires_grad_(True)
    y = L.WeightedSmoothL1Loss(sigma=3.0, code_weights=[1.0] * 7, codewise=True, loss_weight=2.0)(pr, tgt, weights=w.clone())
    y.sum().backward()
    out.update(sl1_pred=pred.numpy(), sl1_tgt=tgt.numpy(), sl1_out=y.detach().numpy(), sl1_grad=pr.grad.numpy())
    # direction softmax: logits (B,A,2), one-hot (B,A,2)
    dl = torch.randn(B, A, 2, generator=g)
    dt = torch.nn.functional.one_hot((torch.rand(B, A, generator=g) < 0.5).long(), 2).float()
    dr = dl.clone().requires_grad_(True)
    y = L.WeightedSoftmaxClassificationLoss(name="direction_classifier", loss_weight=0.2)(dr, dt, weights=w.clone())
    y.sum().backward()
    out.update(dir_logits=dl.numpy(), dir_tgt=dt.numpy(), dir_out=y.detach().numpy(), dir_grad=dr.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "losses_ref.npz"), **out)
    print("losses golden written", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
