"""HIP sparse convolution vs the CPU oracle (oracle/sparse_conv.py, spconv-v1 semantics restated).

Index work is exact: the output SITE SET and every rulebook entry must match (row order is an
implementation choice -- compared as sets / through the dense tensor). Features are float32 sums in
a different order than the oracle's gather-mm-scatter: tolerance 1e-4 relative to the layer's scale."""
import numpy as np
import pytest
import torch

import oracle
from oracle import sparse_conv as osc
from sessd_hip import ops, synth

pytestmark = pytest.mark.gpu


def _random_sites(rng, B, shape, n):
    s = set()
    while len(s) < n:
        s.add((rng.randint(B), rng.randint(shape[0]), rng.randint(shape[1]), rng.randint(shape[2])))
    idx = np.array(sorted(s), np.int32)
    rng.shuffle(idx)
    return idx


def _pad_rows(a, cap):
    out = np.zeros((cap,) + a.shape[1:], a.dtype)
    out[:a.shape[0]] = a
    return out


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (32, 32), (64, 64), (16, 32), (32, 64)])
@pytest.mark.parametrize("ks,st,pd,subm", [(3, 1, 0, True), (3, 2, 1, False), (3, 2, [0, 1, 1], False), ((3, 1, 1), (2, 1, 1), 0, False)])
def test_single_layer(dev, cin, cout, ks, st, pd, subm):
    rng = np.random.RandomState(cin * 7 + cout)
    B, shape, n = 2, [11, 40, 36], 1500
    idx = _random_sites(rng, B, shape, n)
    cap = 2048
    feat = torch.randn(n, cin, generator=torch.Generator().manual_seed(1))
    k3 = osc._triple(ks)
    w = torch.randn(*k3, cin, cout, generator=torch.Generator().manual_seed(2)) * 0.2
    scale = torch.rand(cout) + 0.5
    shift = torch.randn(cout) * 0.1
    want, oidx, oshape, rb = osc.sparse_conv(feat, idx, shape, w, ks, st, pd, subm)
    want = torch.relu(want * scale + shift)

    d_idx = torch.from_numpy(_pad_rows(idx, cap)).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    d_feat = torch.from_numpy(_pad_rows(feat.numpy(), cap)).to(dev)
    in_hash = ops.sparse_hash_build(d_idx, n_dev, shape)
    if subm:
        out_idx, n_out, p3 = d_idx, n_dev, [k // 2 for k in k3]
        nbr, tm = ops.sparse_rulebook(out_idx, n_out, ks, 1, p3, in_hash)
    else:
        out_idx, n_out, out_hash, err = ops.sparse_downsample_sites(d_idx, n_dev, ks, st, pd, oshape, 4096)
        assert int(err.item()) == 0
        nbr, tm = ops.sparse_rulebook(out_idx, n_out, ks, st, pd, in_hash)
    m = int(n_out.item())
    assert m == oidx.shape[0]
    got_idx = out_idx[:m].cpu().numpy()
    # identical site sets
    key = lambda a: set(map(tuple, a.tolist()))
    assert key(got_idx) == key(oidx)
    wpk = ops.sparse_pack_weight(w.to(dev))
    out = ops.sparse_conv(d_feat, nbr, tm, n_out, wpk, cin, cout, scale.to(dev), shift.to(dev), True)
    got = out[:m].cpu()
    # align rows through the coordinates
    lut = {tuple(c): r for r, c in enumerate(oidx.tolist())}
    perm = np.array([lut[tuple(c)] for c in got_idx.tolist()])
    ref = want[perm]
    tol = 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) < tol
    # rulebook entries: number of (in,out) pairs per offset must equal the oracle's
    nb = nbr[:, :m].cpu().numpy()
    for k, (ri, ro) in enumerate(rb[2]):
        assert int((nb[k] >= 0).sum()) == len(ri)


def test_dense_scatter_and_overflow_flag(dev):
    rng = np.random.RandomState(3)
    B, shape, n = 2, [5, 24, 20], 700
    idx = _random_sites(rng, B, shape, n)
    cap = 1024
    feat = torch.randn(n, 64)
    w = torch.randn(3, 1, 1, 64, 64) * 0.1
    want, oidx, oshape, rb = osc.sparse_conv(feat, idx, shape, w, (3, 1, 1), (2, 1, 1), 0, False)
    wd = osc.dense(torch.relu(want), oidx, oshape, B)
    wd = wd.view(B, 64 * oshape[0], oshape[1], oshape[2])
    d_idx = torch.from_numpy(_pad_rows(idx, cap)).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    d_feat = torch.from_numpy(_pad_rows(feat.numpy(), cap)).to(dev)
    in_hash = ops.sparse_hash_build(d_idx, n_dev, shape)
    out_idx, n_out, out_hash, err = ops.sparse_downsample_sites(d_idx, n_dev, (3, 1, 1), (2, 1, 1), 0, oshape, 2048)
    nbr, tm = ops.sparse_rulebook(out_idx, n_out, (3, 1, 1), (2, 1, 1), 0, in_hash)
    dense = torch.zeros((B, 64 * oshape[0], oshape[1], oshape[2]), device=dev)
    ops.sparse_conv(d_feat, nbr, tm, n_out, ops.sparse_pack_weight(w.to(dev)), 64, 64, None, None, True,
                    dense_out=dense, out_indices=out_idx, dense_dims=oshape)
    assert float((dense.cpu() - wd).abs().max()) < 1e-4 * max(1.0, float(wd.abs().max()))
    # capacity overflow is flagged, never silent
    _, n_small, _, err2 = ops.sparse_downsample_sites(d_idx, n_dev, (3, 1, 1), (2, 1, 1), 0, oshape, 64)
    assert int(err2.item()) == 1 and int(n_small.item()) == 64


def test_empty_input(dev):
    cap = 256
    d_idx = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    n_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
    h = ops.sparse_hash_build(d_idx, n_dev, [11, 40, 36])
    out_idx, n_out, out_hash, err = ops.sparse_downsample_sites(d_idx, n_dev, 3, 2, 1, [6, 20, 18], 256)
    assert int(n_out.item()) == 0 and int(err.item()) == 0
    nbr, tm = ops.sparse_rulebook(out_idx, n_out, 3, 2, 1, h)
    out = ops.sparse_conv(torch.zeros(cap, 16, device=dev), nbr, tm, n_out,
                          ops.sparse_pack_weight(torch.zeros(3, 3, 3, 16, 32, device=dev)), 16, 32)
    torch.cuda.synchronize()


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)])
@pytest.mark.parametrize("n,cap", [(1500, 2048), (1501, 1501), (7, 64), (16, 16)])
def test_every_tuning_is_bit_identical(dev, cin, cout, n, cap):
    """cout split x operand depth only tune the launch (sparse_conv.hip): every combination must give the SAME bits
    (fixed (offset, cin) accumulation order), incl. partial last tiles, tiny levels, strided rulebooks and the dense output."""
    rng = np.random.RandomState(cin + cout + n)
    B, shape = 2, [9, 24, 20]
    idx = _random_sites(rng, B, shape, n)
    d_idx = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    d_idx[:n] = torch.from_numpy(idx).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    feat = torch.zeros((cap, cin), device=dev)
    feat[:n] = torch.randn(n, cin, generator=torch.Generator().manual_seed(1)).to(dev)
    w = (torch.randn(3, 3, 3, cin, cout, generator=torch.Generator().manual_seed(2)) * 0.2).to(dev)
    wpk = ops.sparse_pack_weight(w)
    scale, shift = (torch.rand(cout) + 0.5).to(dev), (torch.randn(cout) * 0.1).to(dev)
    h = ops.sparse_hash_build(d_idx, n_dev, shape)
    nbr, tm = ops.sparse_rulebook(d_idx, n_dev, 3, 1, 1, h)
    ref = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=1, depth=2).clone()
    assert float(ref[:n].abs().max()) > 0
    want, _, _, _ = osc.sparse_conv(feat[:n].cpu(), idx, shape, w.cpu(), 3, 1, 0, True)
    want = torch.relu(want * scale.cpu() + shift.cpu())
    assert float((ref[:n].cpu() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    out_idx, n_out, out_hash, err = ops.sparse_downsample_sites(d_idx, n_dev, 3, 2, 1, [5, 12, 10], 4096)
    nbr2, tm2 = ops.sparse_rulebook(out_idx, n_out, 3, 2, 1, h)
    m = int(n_out.item())
    ref2 = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False, cout_split=1, depth=2).clone()
    dref = torch.zeros((B, cout * 5, 12, 10), device=dev)
    ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, True, dense_out=dref, out_indices=out_idx,
                    dense_dims=[5, 12, 10], cout_split=1, depth=2)
    assert m > 0 and float(dref.abs().max()) > 0
    for split in (0, 1, 2, 4):
        if split > 1 and (cout // 16) % split:
            continue
        for depth in (0, 2, 3, 4):
            a = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=split, depth=depth)
            assert torch.equal(a[:n], ref[:n]), (cin, cout, split, depth)
            b = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False, cout_split=split, depth=depth)
            assert torch.equal(b[:m], ref2[:m]), (cin, cout, split, depth)
            d = torch.zeros_like(dref)
            ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, True, dense_out=d, out_indices=out_idx,
                            dense_dims=[5, 12, 10], cout_split=split, depth=depth)
            assert torch.equal(d, dref), (cin, cout, split, depth)
    # two / four tiles per wave (the next tile's neighbour rows fetched under the current tile's MFMAs): the plain kernel's bits
    for tpw in (2, 4):
        for split in (0, 1, 2, 4):
            if split > 1 and (cout // 16) % split:
                continue
            for depth in (0, 2, 3):
                a = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=split, depth=depth, tiles_per_wave=tpw)
                assert torch.equal(a[:n], ref[:n]), (cin, cout, split, depth, tpw)
                b = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False, cout_split=split, depth=depth, tiles_per_wave=tpw)
                assert torch.equal(b[:m], ref2[:m]), (cin, cout, split, depth, tpw)
    # shared-W variant (the four tiles of a workgroup walk the union of their offsets): the plain kernel's bits
    for split in (0, 1, 2, 4):
        if split > 1 and (cout // 16) % split:
            continue
        a = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=split, share_w=1)
        assert torch.equal(a[:n], ref[:n]), (cin, cout, split, "share_w")
        b = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False, cout_split=split, share_w=1)
        assert torch.equal(b[:m], ref2[:m]), (cin, cout, split, "share_w")
    # offset split (four waves share a tile, offsets dealt out by k % 4): one summation order of its own -- the same bits for
    # every cout split / depth, the oracle's tolerance against the unsplit chain; with dense_out the flag is ignored
    kref = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=1, depth=2, offset_split=1).clone()
    kref2 = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False, cout_split=1, depth=2, offset_split=1).clone()
    assert float((kref[:n].cpu() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    assert float((kref2[:m] - ref2[:m]).abs().max()) < 1e-4 * max(1.0, float(ref2[:m].abs().max()))
    for split in (0, 1, 2, 4):
        if split > 1 and (cout // 16) % split:
            continue
        for depth in (0, 2, 3, 4):
            a = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=split, depth=depth, offset_split=1)
            assert torch.equal(a[:n], kref[:n]), (cin, cout, split, depth)
            b = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False, cout_split=split, depth=depth, offset_split=1)
            assert torch.equal(b[:m], kref2[:m]), (cin, cout, split, depth)
    d = torch.zeros_like(dref)
    ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, True, dense_out=d, out_indices=out_idx,
                    dense_dims=[5, 12, 10], cout_split=1, depth=2, offset_split=1)
    assert torch.equal(d, dref)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (64, 64)])
@pytest.mark.parametrize("n", [5000, 1501, 300, 7])
def test_offset_pattern_tiles_give_the_same_bits(dev, cin, cout, n):
    """sessd_sparse_conv_sorted with the chain's perm / tile_mask_sorted (sites of every 256-row group grouped into tiles by their
    neighbour pattern; rows keep their numbers) against the plain tiles on the same tables: BIT-IDENTICAL for the submanifold
    table, the strided table and the dense-output layer, in the plain kernel, the offset split and the shared-W variant -- and
    fewer executed (tile, offset) steps."""
    from oracle import sparse_conv as osc_
    rng = np.random.RandomState(cin + n)
    B, shape0 = 2, [9, 48, 40]
    steps = [(3, 2, 1)]
    idx = _random_sites(rng, B, [8, 48, 40], n)
    cap0 = (n + 63) // 64 * 64 + 64
    d_idx = torch.zeros((cap0, 4), dtype=torch.int32, device=dev)
    d_idx[:n] = torch.from_numpy(idx).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    h0 = ops.sparse_hash_build(d_idx, n_dev, [8, 48, 40])
    oshape = osc_.out_spatial(shape0, 3, 2, 1)
    cap1 = min(8 * cap0, B * int(np.prod(oshape)))
    jobs = [(0, 0, 3, 1, 1), (0, 1, 3, 2, 1)]
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    ch = ops.SparseChain(shape0, steps, [cap1], B, jobs, dev)
    ch.sort_tiles = True
    ch.run(d_idx, n_dev.data_ptr(), cap0, h0, err)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    feat = torch.zeros((cap0, cin), device=dev)
    feat[:n] = torch.randn(n, cin, generator=torch.Generator().manual_seed(1)).to(dev)
    w = (torch.randn(3, 3, 3, cin, cout, generator=torch.Generator().manual_seed(2)) * 0.2).to(dev)
    wpk = ops.sparse_pack_weight(w)
    scale, shift = (torch.rand(cout) + 0.5).to(dev), (torch.randn(cout) * 0.1).to(dev)
    n1 = ch.n_dev[0]
    m = int(n1.item())
    for j, (nd, rows) in enumerate(((n_dev, n), (n1, m))):
        nbr, tm, tms, perm = ch.nbr[j], ch.tile_mask[j], ch.tile_mask_sorted[j], ch.perm[j]
        kv = nbr.shape[0]
        plain_steps = sum(int(((tm[:(rows + 15) // 16] >> k) & 1).sum()) for k in range(kv))
        sorted_steps = sum(int(((tms >> k) & 1).sum()) for k in range(kv))
        assert sorted_steps <= plain_steps
        for kw in (dict(), dict(offset_split=1), dict(share_w=1), dict(cout_split=1, depth=2), dict(cout_split=2, depth=4)):
            a = ops.sparse_conv(feat, nbr, tm, nd, wpk, cin, cout, scale, shift, True, **kw)
            b = ops.sparse_conv(feat, nbr, tms, nd, wpk, cin, cout, scale, shift, True, perm=perm, **kw)
            assert torch.equal(a[:rows], b[:rows]), (j, kw)
        assert float(a[:rows].abs().max()) > 0 or rows == 0
    # the dense-output layer on the strided table
    da, db = torch.zeros((B, cout * oshape[0], oshape[1], oshape[2]), device=dev), torch.zeros((B, cout * oshape[0], oshape[1], oshape[2]), device=dev)
    ops.sparse_conv(feat, ch.nbr[1], ch.tile_mask[1], n1, wpk, cin, cout, scale, shift, True, dense_out=da, out_indices=ch.indices[0], dense_dims=oshape)
    ops.sparse_conv(feat, ch.nbr[1], ch.tile_mask_sorted[1], n1, wpk, cin, cout, scale, shift, True, dense_out=db, out_indices=ch.indices[0],
                    dense_dims=oshape, perm=ch.perm[1])
    assert torch.equal(da, db) and float(da.abs().max()) > 0
