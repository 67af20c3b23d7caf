"""Backward of the sparse convolution: HIP dgrad / wgrad behind the spconv-mirror modules vs torch
autograd through the CPU oracle (oracle/sparse_conv.py: gather -> mm -> index_add, differentiable as written).

Tolerance: gradients are float32 sums over up to thousands of rulebook pairs in a different order than the oracle's:
2e-4 relative to the largest magnitude of the compared tensor."""
import numpy as np
import pytest
import torch

import spconv
from oracle import sparse_conv as osc
from sessd_hip import ops

pytestmark = pytest.mark.gpu


def _random_sites(rng, B, shape, n):
    s = set()
    while len(s) < n:
        s.add((rng.randint(B), rng.randint(shape[0]), rng.randint(shape[1]), rng.randint(shape[2])))
    idx = np.array(sorted(s), np.int32)
    rng.shuffle(idx)
    return idx


def _close(got, want, what):
    scale = max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= 2e-4 * scale, (what, err, scale)


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)])
@pytest.mark.parametrize("ks,st,pd,subm", [(3, 1, 0, True), (3, 2, 1, False), ((3, 1, 1), (2, 1, 1), 0, False)])
def test_layer_gradients(dev, cin, cout, ks, st, pd, subm):
    rng = np.random.RandomState(cin * 5 + cout)
    B, shape, n = 2, [11, 40, 36], 1500
    idx = _random_sites(rng, B, shape, n)
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(n, cin, generator=g)
    k3 = osc._triple(ks)
    w = torch.randn(*k3, cin, cout, generator=g) * 0.2
    # ---- oracle: autograd on CPU
    f_ref = feat.clone().requires_grad_(True)
    w_ref = w.clone().requires_grad_(True)
    y_ref, oidx, oshape, rb = osc.sparse_conv(f_ref, idx, shape, w_ref, ks, st, pd, subm)
    gy_ref = torch.randn(y_ref.shape, generator=g)
    (y_ref * gy_ref).sum().backward()
    # ---- HIP modules
    cls = spconv.SubMConv3d if subm else spconv.SparseConv3d
    conv = cls(cin, cout, ks, st, pd, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w.to(dev))
    f_dev = feat.to(dev).requires_grad_(cin != 4)  # the first layer's input (voxel means) needs no gradient
    x = spconv.SparseConvTensor(f_dev, torch.from_numpy(idx).to(dev), shape, B)
    y = conv(x)
    # match output rows by coordinate (row order of a strided conv is an implementation choice)
    got_idx = y.indices.cpu().numpy().astype(np.int64)
    key = lambda a: ((a[:, 0] * oshape[0] + a[:, 1]) * oshape[1] + a[:, 2]) * oshape[2] + a[:, 3]
    order_ref = np.argsort(key(oidx.astype(np.int64)))
    order_got = np.argsort(key(got_idx))
    assert np.array_equal(key(oidx.astype(np.int64))[order_ref], key(got_idx)[order_got])
    perm = np.empty(len(order_got), np.int64)
    perm[order_got] = order_ref  # row r of ours == row perm[r] of the oracle
    _close(y.features.detach().cpu(), y_ref.detach()[perm], "forward")
    gy = gy_ref[perm].to(dev)
    (y.features * gy).sum().backward()
    _close(conv.weight.grad.cpu(), w_ref.grad, "weight grad")
    if cin != 4:
        _close(f_dev.grad.cpu(), f_ref.grad, "input grad")


def test_transpose_is_inverse_rulebook(dev):
    rng = np.random.RandomState(5)
    B, shape, n = 2, [9, 30, 28], 1200
    idx = _random_sites(rng, B, shape, n)
    d_idx = torch.from_numpy(idx).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    h = ops.sparse_hash_build(d_idx, n_dev, shape)
    oshape = osc.out_spatial(shape, 3, 2, 1)
    oidx, n_out, oh, err = ops.sparse_downsample_sites(d_idx, n_dev, 3, 2, 1, oshape, 4096)
    nbr, tm = ops.sparse_rulebook(oidx, n_out, 3, 2, 1, h)
    nbr_t, tm_t = ops.sparse_rulebook_transpose(nbr, n_out, n)
    a, t = nbr.cpu().numpy(), nbr_t.cpu().numpy()
    m = int(n_out.item())
    pairs = {(k, int(a[k, j]), j) for k in range(27) for j in range(m) if a[k, j] >= 0}
    pairs_t = {(k, i, int(t[k, i])) for k in range(27) for i in range(n) if t[k, i] >= 0}
    assert pairs == pairs_t and len(pairs) > n
    tmt = tm_t.cpu().numpy().astype(np.uint32)
    for tile in range((n + 15) // 16):
        want = 0
        for k in range(27):
            if (t[k, tile * 16:(tile + 1) * 16] >= 0).any():
                want |= 1 << k
        assert int(tmt[tile]) == want


def test_spmiddle_backward_runs_and_matches_oracle(dev):
    """Two stacked layers (subm + strided) with BatchNorm1d(train) and ReLU between them, as SpMiddleFHD stacks them:
    gradients of the first layer's weight flow through dgrad of the second."""
    rng = np.random.RandomState(9)
    B, shape, n = 2, [11, 40, 36], 1200
    idx = _random_sites(rng, B, shape, n)
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(n, 16, generator=g)
    w1 = torch.randn(3, 3, 3, 16, 32, generator=g) * 0.1
    w2 = torch.randn(3, 3, 3, 32, 64, generator=g) * 0.1
    # oracle
    w1r, w2r = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(32, eps=1e-3, momentum=0.01)
    y1, _, _, _ = osc.sparse_conv(feat, idx, shape, w1r, 3, 1, 0, True)
    y1 = torch.relu(bn(y1))
    y2, oidx, oshape, _ = osc.sparse_conv(y1, idx, shape, w2r, 3, 2, 1, False)
    y2.pow(2).sum().backward()
    # HIP
    net = spconv.SparseSequential(spconv.SubMConv3d(16, 32, 3, bias=False, indice_key="s"), torch.nn.BatchNorm1d(32, eps=1e-3, momentum=0.01),
                                  torch.nn.ReLU(), spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False)).to(dev)
    with torch.no_grad():
        net[0].weight.copy_(w1.to(dev))
        net[3].weight.copy_(w2.to(dev))
    out = net(spconv.SparseConvTensor(feat.to(dev), torch.from_numpy(idx).to(dev), shape, B))
    out.features.pow(2).sum().backward()
    _close(net[3].weight.grad.cpu(), w2r.grad, "w2 grad")
    _close(net[0].weight.grad.cpu(), w1r.grad, "w1 grad")


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 64), (64, 64), (64, 128)])
@pytest.mark.parametrize("reverse", [False, True])
def test_adjoint_weight_pack_equals_flip_transpose_pack(dev, cin, cout, reverse):
    """sessd_sparse_pack_weight_adjoint (the data-gradient conv's packed weight straight from the layer's weight) == packing the
    flipped / transposed copy: identical bits."""
    import torch
    from sessd_hip import ops
    w = torch.randn(3, 3, 3, cin, cout, generator=torch.Generator().manual_seed(cin + cout)).to(dev)
    ref = w.reshape(27, cin, cout)
    if reverse:
        ref = ref.flip(0)
    want = ops.sparse_pack_weight(ref.transpose(1, 2).contiguous())
    got = ops.sparse_pack_weight_adjoint(w, reverse)
    assert torch.equal(got, want)
