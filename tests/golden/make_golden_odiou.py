"""Generates tests/golden/odiou_ref.npz by running the REFERENCE's ODIoU loss from source on CPU:
    det3d/models/losses/odious.py  (odiou_3D :837-900 with compute_vertex / sort_vertex / area_polygon / mbr_diag_compute)
Every pair is evaluated alone (weights = 1, batch_size = 1) so that the per-pair term (= loss / 2) and its gradient with
respect to the predicted box are recorded; one batched call records the weighted / normalised aggregate.
Run in the build container only (needs /root/reference):  python tests/golden/make_golden_odiou.py"""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_ref(relpath, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def make_pairs(seed=0):
    rng = np.random.RandomState(seed)
    n = 160
    g = np.zeros((n, 7), np.float32)
    g[:, 0] = rng.uniform(0, 60, n)
    g[:, 1] = rng.uniform(-30, 30, n)
    g[:, 2] = rng.uniform(-2, 0, n)
    g[:, 3] = rng.uniform(1.4, 1.9, n)
    g[:, 4] = rng.uniform(3.2, 4.6, n)
    g[:, 5] = rng.uniform(1.3, 1.8, n)
    g[:, 6] = rng.uniform(-3.1, 3.1, n)
    q = g.copy()
    # graded perturbations: tiny, moderate, large, disjoint
    scale = np.repeat(np.array([0.02, 0.15, 0.5, 1.5], np.float32), n // 4)[:, None]
    q += rng.normal(0, 1, (n, 7)).astype(np.float32) * scale * np.array([1, 1, 0.5, 0.3, 0.5, 0.3, 0.6], np.float32)
    q[:, 3:6] = np.abs(q[:, 3:6]) + 0.05
    # special cases
    q[0] = g[0] + np.array([30, 30, 0, 0, 0, 0, 0], np.float32)          # far apart (no intersection)
    q[1] = g[1]; q[1, 6] += np.float32(np.pi / 2)                          # same centre, rotated 90 degrees
    q[2] = g[2]; q[2, 3:5] *= 0.5                                         # contained
    q[3] = g[3]; q[3, 2] += 5.0                                           # no height overlap
    q[4] = g[4]; q[4, 4] = -1.0                                           # invalid predicted size: indicator false
    q[5] = g[5]; q[5, 0] = 250.0                                          # clamped coordinate
    q[6] = g[6] + np.array([0.3, -0.2, 0.1, 0.05, -0.1, 0.02, 0.0], np.float32); q[6, 6] = g[6, 6] + 3.0
    return g, q.astype(np.float32)


def main():
    assert os.path.isdir(REF)
    warnings.filterwarnings("ignore")
    od = load_ref("det3d/models/losses/odious.py", "ref_odious")
    g, q = make_pairs()
    n = g.shape[0]
    term = np.zeros(n, np.float64)
    grad = np.zeros((n, 7), np.float64)
    loss_fn = od.odiou_3D()
    for i in range(n):
        gt = torch.from_numpy(g[i:i + 1])
        qt = torch.from_numpy(q[i:i + 1]).clone().requires_grad_(True)
        loss = loss_fn(gt, qt, torch.ones(1), 1)
        loss.backward()
        term[i] = float(loss.detach()) / 2.0
        grad[i] = qt.grad[0].numpy().astype(np.float64) / 2.0
    rng = np.random.RandomState(7)
    w = rng.uniform(0, 2, n).astype(np.float32)
    # the batched form only works when every row is valid (odious.py:897 adds the UNFILTERED angle term to the filtered
    # ones: a shape error otherwise) -- which holds in training, where sizes are exp-decoded
    valid = np.all(g[:, 3:6] > 0, 1) & np.all(q[:, 3:6] > 0, 1)
    qt = torch.from_numpy(q[valid]).clone().requires_grad_(True)
    loss = loss_fn(torch.from_numpy(g[valid]), qt, torch.from_numpy(w[valid]), 4)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "odiou_ref.npz"), g=g, q=q, term=term, grad=grad, weights=w, valid=valid,
                        batch_loss=float(loss.detach()), batch_grad=qt.grad.numpy())
    print("odiou golden written: terms %.4f .. %.4f, batch loss %.5f, nonfinite grads %d" %
          (term.min(), term.max(), float(loss.detach()), int((~np.isfinite(grad)).sum())))


if __name__ == "__main__":
    main()
