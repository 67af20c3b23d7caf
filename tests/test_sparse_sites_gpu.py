"""Whole-chain site generation + rulebooks (csrc/sparse_sites.hip, ops.SparseChain) vs the CPU oracle's per-layer rulebook
(oracle/sparse_conv.py, spconv-v1 semantics pinned against F.conv3d in tests/test_oracle_sparse_conv_cpu.py).

Index work is exact: per level the site SET must equal the oracle's, rows must come in ascending (b,z,y,x) order, the live
count must match; every neighbour table must hold exactly the oracle's (input cell, output cell) pairs per kernel offset, and
the tile masks must be the OR over each 16-site tile. Geometry = the four SparseConv3d of SpMiddleFHD (scn.py:113,122,134,146)."""
import numpy as np
import pytest
import torch

from oracle import sparse_conv as osc
from sessd_hip import ops

pytestmark = pytest.mark.gpu

STEPS = [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1]), ((3, 1, 1), (2, 1, 1), 0)]


def _sites(rng, B, shape, n, clustered):
    if clustered:  # surfaces: a few planes with holes, like a lidar scan (dense neighbourhoods)
        pts = set()
        while len(pts) < n:
            b = rng.randint(B)
            z0 = rng.randint(shape[0])
            y0, x0 = rng.randint(shape[1]), rng.randint(shape[2])
            for _ in range(60):
                y, x = y0 + rng.randint(-6, 7), x0 + rng.randint(-6, 7)
                z = z0 + rng.randint(0, 2)
                if 0 <= z < shape[0] and 0 <= y < shape[1] and 0 <= x < shape[2]:
                    pts.add((b, z, y, x))
        idx = np.array(sorted(pts), np.int32)[:n]
    else:
        cells = B * shape[0] * shape[1] * shape[2]
        lin = rng.choice(cells, size=n, replace=False)
        idx = np.stack([lin // (shape[0] * shape[1] * shape[2]), (lin // (shape[1] * shape[2])) % shape[0],
                        (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
    rng.shuffle(idx)
    return idx


def _oracle_chain(idx, shape0):
    levels = [(idx, list(shape0))]
    for ks, st, pd in STEPS:
        oidx, oshape, _ = osc.rulebook(levels[-1][0], levels[-1][1], ks, st, pd, False)
        levels.append((oidx, oshape))
    return levels


def _pairs_from_table(nbr_k, in_idx, out_idx, m):
    rows = np.nonzero(nbr_k[:m] >= 0)[0]
    return set(map(tuple, np.concatenate([in_idx[nbr_k[rows]], out_idx[rows]], 1).tolist()))


@pytest.mark.parametrize("B,shape0,n,clustered,seed", [
    (2, [41, 64, 56], 3000, True, 0),
    (1, [41, 48, 40], 1, False, 1),       # a single voxel
    (3, [41, 32, 40], 2500, False, 2),    # isolated voxels: ~8 outputs per input at level 1
    (1, [41, 200, 176], 9000, True, 3),   # W = 176 is not a multiple of 32: grid rows straddle occupancy words
    (2, [41, 40, 32], 0, False, 4),       # empty input
])
def test_chain_sites_and_rulebooks(dev, B, shape0, n, clustered, seed):
    rng = np.random.RandomState(seed)
    idx = _sites(rng, B, [shape0[0] - 1, shape0[1], shape0[2]], n, clustered) if n else np.zeros((0, 4), np.int32)
    n = idx.shape[0]
    cap0 = max(64, (n + 63) // 64 * 64 + 64)
    d_idx = torch.zeros((cap0, 4), dtype=torch.int32, device=dev)
    d_idx[:n] = torch.from_numpy(idx).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    hash_dims = [shape0[0] - 1, shape0[1], shape0[2]]  # the voxelizer's key dims (z < 40), scn.py:179 adds the 41st plane
    h0 = ops.sparse_hash_build(d_idx, n_dev, hash_dims)
    want = _oracle_chain(idx, shape0)
    caps = [max(64, w[0].shape[0] + 37) for w in want[1:]]
    jobs = []
    for l in range(4):
        jobs.append((l, l, 3, 1, 1))                      # submanifold table on level l
        jobs.append((l, l + 1) + tuple(STEPS[l]))          # strided table into level l + 1
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    ch = ops.SparseChain(shape0, STEPS, caps, B, jobs, dev)
    ch.sort_tiles = True     # offset-pattern tiles next to the plain tables (checked at the end)
    ch.run(d_idx, n_dev.data_ptr(), cap0, h0, err)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    got_levels = [idx]
    for l in range(4):
        m = int(ch.n_dev[l].item())
        oidx, oshape = want[l + 1]
        assert ch.shapes[l + 1] == list(oshape)
        assert m == oidx.shape[0], (l, m, oidx.shape[0])
        g = ch.indices[l][:m].cpu().numpy()
        lin = osc._lin(g, oshape)
        assert np.all(np.diff(lin) > 0), "rows must be in ascending (b,z,y,x) order"
        assert np.array_equal(np.sort(lin), np.sort(osc._lin(oidx, oshape)))
        got_levels.append(g)
    # neighbour tables
    for j, (li, lo, ks, st, pd) in enumerate(jobs):
        subm = li == lo
        in_idx, out_idx = got_levels[li], got_levels[lo]
        m = out_idx.shape[0]
        shape_in = want[li][1]
        ref_out, _, pairs = osc.rulebook(in_idx, shape_in, ks, st, pd, subm)
        nbr = ch.nbr[j].cpu().numpy()
        tm = ch.tile_mask[j].cpu().numpy().view(np.uint32)
        kv = nbr.shape[0]
        assert kv == len(pairs)
        for k, (ri, ro) in enumerate(pairs):
            ref = set(map(tuple, np.concatenate([in_idx[ri], ref_out[ro]], 1).tolist())) if len(ri) else set()
            assert _pairs_from_table(nbr[k], in_idx, out_idx, m) == ref, (j, k)
        for t in range((m + 15) // 16):
            hit = (nbr[:, t * 16:min(m, t * 16 + 16)] >= 0).any(1)
            assert int(tm[t]) == sum(1 << k for k in range(kv) if hit[k]), (j, t)
        assert not tm[(m + 15) // 16:].any()
        # ---- offset-pattern tiles (sessd_rulebook_job_t.site_mask / perm / tile_mask_sorted)
        sm = ch.site_mask[j].cpu().numpy().view(np.uint32)[:m]
        want_sm = np.zeros(m, np.uint32)
        for k in range(kv):
            want_sm |= ((nbr[k, :m] >= 0).astype(np.uint32) << np.uint32(k))
        assert np.array_equal(sm, want_sm), j
        perm = ch.perm[j].cpu().numpy()
        tms = ch.tile_mask_sorted[j].cpu().numpy().view(np.uint32)
        for g in range((m + 255) // 256):
            live = min(256, m - g * 256)
            p = perm[g * 256:g * 256 + 256].astype(np.int64)
            assert sorted(p.tolist()) == list(range(256)), (j, g)            # a permutation of the group's 256 rows
            assert set(p[:live].tolist()) == set(range(live))                 # the live rows first ...
            keys = (sm[g * 256 + p[:live]].astype(np.int64) << 8) | p[:live]
            assert np.all(np.diff(keys) > 0), (j, g)                          # ... ordered by (pattern, row)
            for t in range(16):
                rows = p[t * 16:min(live, t * 16 + 16)]
                want_t = np.bitwise_or.reduce(sm[g * 256 + rows]) if len(rows) else 0
                assert int(tms[g * 16 + t]) == int(want_t), (j, g, t)
        assert not tms[((m + 255) // 256) * 16:].any()


def test_capacity_overflow_is_flagged(dev):
    rng = np.random.RandomState(9)
    shape0, B = [41, 32, 40], 1
    idx = _sites(rng, B, [40, 32, 40], 1500, False)
    d_idx = torch.from_numpy(idx).to(dev)
    n_dev = torch.tensor([1500], dtype=torch.int32, device=dev)
    h0 = ops.sparse_hash_build(d_idx, n_dev, [40, 32, 40])
    want = _oracle_chain(idx, shape0)
    caps = [want[1][0].shape[0] - 100] + [w[0].shape[0] + 8 for w in want[2:]]
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    ch = ops.SparseChain(shape0, STEPS, caps, B, [(1, 1, 3, 1, 1)], dev)
    ch.run(d_idx, n_dev.data_ptr(), 1500, h0, err)
    torch.cuda.synchronize()
    assert int(err.item()) == 1 and int(ch.n_dev[0].item()) == caps[0]
    nbr = ch.nbr[0].cpu().numpy()
    assert nbr.max() < caps[0]  # rows beyond the capacity read as absent
    # deeper levels are complete: their cells were marked from level 0 directly
    for l in (1, 2, 3):
        assert int(ch.n_dev[l].item()) == want[l + 1][0].shape[0]


def test_reuse_without_clear_needs_the_callers_fill(dev):
    """clear=False is the engine's mode (one arena fill per frame): two runs with a zero fill in between give identical tables."""
    rng = np.random.RandomState(11)
    shape0 = [41, 48, 40]
    idx = _sites(rng, 1, [40, 48, 40], 2000, True)
    n = idx.shape[0]
    d_idx = torch.from_numpy(idx).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    h0 = ops.sparse_hash_build(d_idx, n_dev, [40, 48, 40])
    want = _oracle_chain(idx, shape0)
    caps = [w[0].shape[0] + 64 for w in want[1:]]
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    ch = ops.SparseChain(shape0, STEPS, caps, 1, [(1, 2) + tuple(STEPS[1])], dev)
    ch.run(d_idx, n_dev.data_ptr(), n, h0, err, clear=True)
    a = (ch.nbr[0].clone(), [i.clone() for i in ch.indices], [int(c.item()) for c in ch.n_dev])
    ch.ws.zero_()
    ch.run(d_idx, n_dev.data_ptr(), n, h0, err, clear=False)
    torch.cuda.synchronize()
    assert torch.equal(a[0], ch.nbr[0]) and a[2] == [int(c.item()) for c in ch.n_dev]
    for x, y, m in zip(a[1], ch.indices, a[2]):
        assert torch.equal(x[:m], y[:m])
