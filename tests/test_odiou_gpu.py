"""sessd_odiou3d (one-launch differentiable ODIoU loss) vs oracle/odiou.py in the kernel's hull
convention (tight: same float64 procedure) and vs the reference's own run (tests/golden/odiou_ref.npz, the tolerance of
tests/test_odiou_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import odiou
from sessd_hip import ops
from test_odiou_cpu import unambiguous_pairs

pytestmark = pytest.mark.gpu


def test_kernel_vs_oracle_and_reference(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "odiou_ref.npz"))
    G, Q = g["g"], g["q"]
    gq = torch.from_numpy(Q).to(dev).requires_grad_(True)
    w = torch.ones(len(G), device=dev)
    loss = ops.odiou_3d_loss(torch.from_numpy(G).to(dev), gq, w, 1)
    loss.backward()
    grad = gq.grad.cpu().numpy() / 2.0  # loss = 2 * sum(term) / 1
    want_t = np.zeros(len(G))
    want_g = np.zeros((len(G), 7))
    for i in range(len(G)):
        want_t[i], want_g[i] = odiou.odiou_term(G[i], Q[i], device_convention=True)
    assert abs(float(loss.detach()) / 2.0 - want_t.sum()) < 1e-4 * want_t.sum()
    assert np.abs(grad - want_g).max() < 2e-5 * max(1.0, np.abs(want_g).max())  # float32 outputs of a float64 computation
    keep = unambiguous_pairs(g)                                                    # the reference itself, where well defined
    assert np.abs(grad[keep] - g["grad"][keep]).max() < 1e-2


def test_weights_batch_size_and_empty(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "odiou_ref.npz"))
    v = g["valid"] & unambiguous_pairs(g)
    w = g["weights"]
    q = torch.from_numpy(g["q"][v]).to(dev).requires_grad_(True)
    loss = ops.odiou_3d_loss(torch.from_numpy(g["g"][v]).to(dev), q, torch.from_numpy(w[v]).to(dev), 4)
    (3.0 * loss).backward()
    want = 2.0 * float((g["term"][v] * w[v]).sum()) / 4          # odious.py:895-899 on the reference's per-pair terms
    assert abs(float(loss.detach()) - want) < 2e-4 * want
    want_g = 2.0 * g["grad"][v] * w[v][:, None] / 4
    assert np.abs(q.grad.cpu().numpy() / 3.0 - want_g).max() < 1e-2
    e = ops.odiou_3d_loss(torch.zeros((0, 7), device=dev), torch.zeros((0, 7), device=dev, requires_grad=True),
                          torch.zeros((0,), device=dev), 2)
    assert float(e) == 0.0
