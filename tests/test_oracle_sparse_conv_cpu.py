"""Pin of oracle/sparse_conv.py (spconv is absent: SURVEY.md 8c) against torch.nn.functional.conv3d on the densified grid.

spconv v1 semantics restated by the oracle (call sites det3d/models/backbones/scn.py:106-148,182-187):
  * SparseConv3d(k, s, p): identical to a dense cross-correlation of the zero-filled grid, restricted to the output cells
    whose receptive field holds at least one active input -> equality must hold EVERYWHERE on the dense output grid
    (cells outside the site set are exactly 0 in the dense result only if no input reaches them), and the site set must
    be exactly the set of reachable cells;
  * SubMConv3d(k): the dense 'same' cross-correlation (pad = k//2) sampled at the ACTIVE input sites only.
Weights are [kz,ky,kx,Cin,Cout] -> conv3d's [Cout,Cin,kz,ky,kx]. Covers the four strided geometries and the six channel
pairs of SpMiddleFHD, the stacked 14-layer network on a small grid, and `.dense()` + view (channel = c*D + z).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sparse_conv as sc

# (ksize, stride, padding) of the four SparseConv3d of SpMiddleFHD (scn.py:113,122,134,146)
STRIDED = [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1]), ((3, 1, 1), (2, 1, 1), 0)]
PAIRS = [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)]


def _sites(rng, B, shape, n):
    cells = B * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=min(n, cells), replace=False)
    x = lin % shape[2]
    y = (lin // shape[2]) % shape[1]
    z = (lin // (shape[2] * shape[1])) % shape[0]
    b = lin // (shape[2] * shape[1] * shape[0])
    return np.stack([b, z, y, x], 1).astype(np.int32)


def _densify(feat, idx, B, shape):
    d = torch.zeros((B, feat.shape[1]) + tuple(shape), dtype=torch.float32)
    i = torch.from_numpy(idx.astype(np.int64))
    d[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] = feat
    return d


def _w_dense(w):
    return w.permute(4, 3, 0, 1, 2).contiguous()  # [kz,ky,kx,Cin,Cout] -> [Cout,Cin,kz,ky,kx]


@pytest.mark.parametrize("cin,cout", PAIRS)
@pytest.mark.parametrize("geom", range(4))
def test_strided_conv_equals_dense_conv3d(cin, cout, geom):
    ks, st, pd = STRIDED[geom]
    rng = np.random.default_rng(100 * geom + cin + cout)
    torch.manual_seed(7 * geom + cin)
    B, shape = 2, [9, 14, 12]
    idx = _sites(rng, B, shape, 300)
    feat = torch.randn(idx.shape[0], cin)
    k3 = sc._triple(ks)
    w = torch.randn(k3[0], k3[1], k3[2], cin, cout) * 0.2
    out, oidx, oshape, rb = sc.sparse_conv(feat, idx, shape, w, ks, st, pd, subm=False)
    ref = F.conv3d(_densify(feat, idx, B, shape), _w_dense(w), None, stride=sc._triple(st), padding=sc._triple(pd))
    assert list(ref.shape[2:]) == list(oshape)
    got = _densify(out, oidx, B, oshape)
    assert torch.allclose(got, ref, rtol=0, atol=2e-5 * float(ref.abs().max()))
    # site set == cells reachable from an active input (occupancy convolved with a ones kernel)
    occ = _densify(torch.ones(idx.shape[0], 1), idx, B, shape)
    reach = F.conv3d(occ, torch.ones(1, 1, *k3), None, stride=sc._triple(st), padding=sc._triple(pd))[:, 0] > 0
    mine = torch.zeros_like(reach)
    oi = torch.from_numpy(oidx.astype(np.int64))
    mine[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]] = True
    assert torch.equal(mine, reach)
    assert len(np.unique(sc._lin(oidx, oshape))) == oidx.shape[0]  # no duplicate rows


@pytest.mark.parametrize("cin,cout", PAIRS)
def test_subm_conv_equals_dense_conv3d_at_active_sites(cin, cout):
    rng = np.random.default_rng(cin * 31 + cout)
    torch.manual_seed(cin + cout)
    B, shape = 2, [7, 16, 13]
    idx = _sites(rng, B, shape, 500)
    feat = torch.randn(idx.shape[0], cin)
    w = torch.randn(3, 3, 3, cin, cout) * 0.2
    for padding in (0, 1):  # SubM ignores `padding` (scn.py builds it both ways, :24-44 vs :107)
        out, oidx, oshape, rb = sc.sparse_conv(feat, idx, shape, w, 3, 1, padding, subm=True)
        assert np.array_equal(oidx, idx) and list(oshape) == shape
        ref = F.conv3d(_densify(feat, idx, B, shape), _w_dense(w), None, stride=1, padding=1)
        i = torch.from_numpy(idx.astype(np.int64))
        want = ref[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]]
        assert torch.allclose(out, want, rtol=0, atol=2e-5 * float(want.abs().max()))


def test_stacked_spmiddle_equals_masked_dense_network():
    """All 14 layers (scn.py:106-148) on a small grid: dense conv3d network with the submanifold masks applied by hand."""
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    B, in_shape = 2, [16, 24, 40]  # x, y, z grid -> sparse shape [41, 24, 16] (z: 41 -> 21 -> 11 -> 5 -> 2 as in the model)
    shape = [in_shape[2] + 1, in_shape[1], in_shape[0]]
    idx = _sites(rng, B, [in_shape[2], in_shape[1], in_shape[0]], 900)
    feat = torch.randn(idx.shape[0], 4)
    ws, bns = [], []
    for (kind, cin, cout, ks, st, pd, key) in sc.SPMIDDLE_FHD_LAYERS:
        k3 = sc._triple(ks)
        ws.append(torch.randn(k3[0], k3[1], k3[2], cin, cout) * (1.5 / np.sqrt(cin * k3[0] * k3[1] * k3[2])))
        bns.append(dict(weight=torch.rand(cout) + 0.5, bias=torch.randn(cout) * 0.1, running_mean=torch.randn(cout) * 0.1,
                        running_var=torch.rand(cout) + 0.5))
    got, levels = sc.spmiddle_fhd(feat, idx, B, in_shape, ws, bns, return_levels=True)
    x = _densify(feat, idx, B, shape)
    mask = _densify(torch.ones(idx.shape[0], 1), idx, B, shape) > 0
    for (kind, cin, cout, ks, st, pd, key), w, bn in zip(sc.SPMIDDLE_FHD_LAYERS, ws, bns):
        k3 = sc._triple(ks)
        if kind == "subm":
            y = F.conv3d(x, _w_dense(w), None, stride=1, padding=1)
        else:
            y = F.conv3d(x, _w_dense(w), None, stride=sc._triple(st), padding=sc._triple(pd))
            mask = F.conv3d(mask.float(), torch.ones(1, 1, *k3), None, stride=sc._triple(st), padding=sc._triple(pd)) > 0
        y = F.batch_norm(y, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-3)
        x = torch.relu(y) * mask  # features exist only at active sites
    N, C, D, H, W = x.shape
    want = x.reshape(N, C * D, H, W)  # scn.py:186-187: channel = c*D + z
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=0, atol=5e-5 * float(want.abs().max()))
    # site counts of every level follow the mask
    assert levels[-1][1].shape[0] == int(mask.sum())


def test_dense_layout_channel_is_c_times_d_plus_z():
    idx = np.array([[0, 1, 2, 3], [1, 0, 0, 0]], np.int32)
    feat = torch.tensor([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]])
    d = sc.dense(feat, idx, [2, 4, 5], 2)
    assert d.shape == (2, 3, 2, 4, 5)
    v = d.view(2, 6, 4, 5)
    for c in range(3):
        assert float(v[0, c * 2 + 1, 2, 3]) == float(feat[0, c])
        assert float(v[1, c * 2 + 0, 0, 0]) == float(feat[1, c])
    assert float(v.abs().sum()) == float(feat.abs().sum())


def test_out_spatial_equals_the_reference_held_shape_annotations():
    """The one pin the reference itself holds for spconv's output-size rule: the shapes its author annotated next to the four
    strided convs (scn.py:113,122,134,146; parsed from source by tests/golden/make_golden_scn_shapes.py). The oracle's rule
    `(D + 2p - k)//s + 1` must reproduce every annotated output shape from the annotated input shape and the constructor
    arguments on that line, the annotations must chain, they must be the geometry the tests above pin (STRIDED), and the
    product's layer table (sessd_hip.engine.SPMIDDLE_LAYERS) must carry the same arguments."""
    import json
    import os
    from sessd_hip.engine import SPMIDDLE_LAYERS
    L = lambda v: [int(v)] * 3 if isinstance(v, int) else [int(x) for x in v]
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scn_shapes.json")))
    rows = g["rows"]
    assert [r["line"] for r in rows] == [113, 122, 134, 146]
    assert rows[0]["in_shape"] == [41, 1600, 1408]          # scn.py:179: input_shape[::-1] + [1, 0, 0]
    ours = [lay for lay in SPMIDDLE_LAYERS if lay[0] == "conv"]
    for r, (ks, st, pd), lay in zip(rows, STRIDED, ours):
        assert (L(r["ksize"]), L(r["stride"]), L(r["padding"])) == (L(ks), L(st), L(pd))
        assert (lay[1], lay[2], L(lay[3]), L(lay[4]), L(lay[5])) == (r["cin"], r["cout"], L(ks), L(st), L(pd))
        assert sc.out_spatial(r["in_shape"], r["ksize"], r["stride"], r["padding"]) == r["out_shape"], r
    for a, b in zip(rows[:-1], rows[1:]):
        assert a["out_shape"] == b["in_shape"]
    assert rows[-1]["out_shape"] == [2, 200, 176]           # scn.py:186-187: view(B, 64 * 2, 200, 176)
