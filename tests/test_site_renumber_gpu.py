"""sessd_sparse_renumber_sites (engine-internal grid-row numbering of the voxels, optional: `InferenceEngine(sort_sites=True)`).

What must hold: the output is a permutation of the input sites with their features, rows are ordered by (batch, z, y), the
level's hash points at the new rows, and the whole engine produces bit-identical detections and BEV map with the option on.
(First run on hardware in round 2: green.)"""
import os

import numpy as np
import pytest
import torch

from sessd_hip import ops

pytestmark = pytest.mark.gpu


def _sites(rng, B, shape, n):
    cells = B * shape[0] * shape[1] * shape[2]
    lin = np.unique(rng.randint(0, cells, size=4 * n + 64)) if cells > 4 * n else np.arange(cells)
    lin = rng.permutation(lin)[:n]
    assert len(lin) == n
    x = lin % shape[2]; y = (lin // shape[2]) % shape[1]; z = (lin // (shape[2] * shape[1])) % shape[0]; b = lin // (shape[2] * shape[1] * shape[0])
    return np.stack([b, z, y, x], 1).astype(np.int32)


@pytest.mark.parametrize("B,shape,n,cap", [(1, [40, 1600, 1408], 15000, 16000), (2, [40, 1600, 1408], 9000, 12032),
                                           (3, [5, 7, 9], 600, 1024), (1, [2, 3, 4], 1, 64)])
def test_permutation_row_order_and_hash(dev, B, shape, n, cap):
    rng = np.random.RandomState(n)
    idx = _sites(rng, B, shape, n)
    feat = rng.randn(n, 4).astype(np.float32)
    d_idx = torch.zeros((cap, 4), dtype=torch.int32, device=dev); d_idx[:n] = torch.from_numpy(idx).to(dev)
    d_feat = torch.zeros((cap, 4), dtype=torch.float32, device=dev); d_feat[:n] = torch.from_numpy(feat).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    h = ops.sparse_hash_build(d_idx, n_dev, shape)
    out_idx, out_feat = ops.sparse_renumber_sites(d_idx, n_dev, d_feat, h, B)
    torch.cuda.synchronize()
    oi, of = out_idx[:n].cpu().numpy(), out_feat[:n].cpu().numpy()
    key = lambda a: ((a[:, 0].astype(np.int64) * shape[0] + a[:, 1]) * shape[1] + a[:, 2]) * shape[2] + a[:, 3]
    # a permutation of the sites, every site with its own features
    order_in, order_out = np.argsort(key(idx)), np.argsort(key(oi))
    assert np.array_equal(idx[order_in], oi[order_out]) and np.array_equal(feat[order_in], of[order_out])
    # numbered by grid row
    rows = key(oi) // shape[2]
    assert np.all(np.diff(rows) >= 0)
    # the hash now answers with the new rows: an identity "convolution" (1x1x1) looks every site up
    nbr, _ = ops.sparse_rulebook(out_idx, n_dev, 1, 1, 0, h)
    assert np.array_equal(nbr[0, :n].cpu().numpy(), np.arange(n, dtype=np.int32))
    # the input tables are untouched
    assert np.array_equal(d_idx[:n].cpu().numpy(), idx)


def test_engine_results_do_not_depend_on_the_numbering(dev):
    from sessd_hip import configs, synth
    from sessd_hip.engine import InferenceEngine
    VG = configs.VOXEL_GENERATOR
    model = configs.build_synthetic_detector(dev, seed=0, max_voxels=16000, num_points=20000)
    mk = lambda flag, B: InferenceEngine(model, VG["range"], VG["voxel_size"], VG["max_points_in_voxel"], 16000, configs.TEST_CFG,
                                         batch_size=B, max_points_per_frame=20000, device=dev, sort_sites=flag)
    for B in (1, 2):
        frames = [torch.from_numpy(synth.make_frame(7 + i, 20000)).to(dev) for i in range(B)]
        outs = []
        for flag in (False, True):
            e = mk(flag, B)
            e.set_points(frames)
            e.enqueue()
            torch.cuda.synchronize()
            outs.append((e.results(), e.bev.clone(), [int(L["n"].item()) for L in e.levels[1:]]))
            if flag:   # and through a captured graph
                e.capture()
                e.set_points(frames)
                e.replay()
                torch.cuda.synchronize()
                again = e.results()
                for a, b in zip(outs[-1][0], again):
                    assert all(np.array_equal(a[k], b[k]) for k in a)
        (r0, bev0, n0), (r1, bev1, n1) = outs
        assert n0 == n1 and torch.equal(bev0, bev1)
        for a, b in zip(r0, r1):
            assert len(a["scores"]) > 0 and all(np.array_equal(a[k], b[k]) for k in a)
