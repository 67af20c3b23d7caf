"""sessd_bn_relu_train_fwd / _bwd (train-mode BatchNorm1d + ReLU over a sparse feature table, csrc/bn_train.hip) vs
torch.nn.BatchNorm1d(eps=1e-3, momentum=0.01) + ReLU on the same rows: output, running statistics, and the gradients with
respect to the input, weight and bias. Tolerances: 2e-5 on values of O(1) (float32, different reduction order)."""
import os

import pytest
import torch

from sessd_hip import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C", [2, 4, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("n,cap", [(15000, 16000), (1, 64), (2, 2), (3001, 3001)])
@pytest.mark.parametrize("relu", [True, False])
def test_matches_torch_batchnorm(dev, C, n, cap, relu):
    g = torch.Generator().manual_seed(C * 1000 + n)
    x = torch.zeros(cap, C)
    x[:n] = torch.randn(n, C, generator=g) * 1.7 + 0.3
    x[n:] = 1e6        # rows past the count must not matter
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    up = torch.randn(cap, C, generator=g)
    ref = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
    mine = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
    ref.weight.data.copy_(w); ref.bias.data.copy_(b)
    ref.running_mean.data.copy_(torch.randn(C, generator=g) * 0.1); ref.running_var.data.copy_(torch.rand(C, generator=g) + 0.5)
    mine.load_state_dict(ref.state_dict())
    xr = x[:n].to(dev).clone().requires_grad_(True)
    if n > 1:
        yr = ref(xr)
        yr = torch.relu(yr) if relu else yr
        (yr * up[:n].to(dev)).sum().backward()
    xm = x.to(dev).clone().requires_grad_(True)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    ym = ops.bn_relu_train(xm, n_dev, mine, relu=relu)
    (ym[:n] * up[:n].to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(ym[:n]).all() and float(ym[n:].abs().sum()) == 0 and float(xm.grad[n:].abs().sum()) == 0
    if n > 1:
        assert torch.allclose(ym[:n], yr, rtol=0, atol=2e-5 * max(1.0, float(yr.abs().max())))
        assert torch.allclose(mine.running_mean, ref.running_mean, rtol=0, atol=1e-6)
        assert torch.allclose(mine.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1
        scale = max(1.0, float(xr.grad.abs().max()))
        assert torch.allclose(xm.grad[:n], xr.grad, rtol=0, atol=5e-5 * scale)
        assert torch.allclose(mine.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(mine.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-3)
    # deterministic: a second run gives the same bits
    mine2 = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
    mine2.weight.data.copy_(w); mine2.bias.data.copy_(b)
    y2 = ops.bn_relu_train(x.to(dev), n_dev, mine2, relu=relu)
    assert torch.equal(y2, ym.detach())


def test_deferred_batch_counts(dev):
    """ops.deferred_batch_counts: the layers' num_batches_tracked are incremented once each, together, when the context closes
    (TrainStep wraps an iteration's forward passes in it); outputs are those of the immediate form."""
    bns = [torch.nn.BatchNorm1d(16, eps=1e-3, momentum=0.01).to(dev).train() for _ in range(3)]
    bn2 = torch.nn.BatchNorm2d(8, eps=1e-3, momentum=0.01).to(dev).train()
    x = torch.randn(500, 16, device=dev)
    n_dev = torch.tensor([500], dtype=torch.int32, device=dev)
    with ops.deferred_batch_counts():
        ys = [ops.bn_relu_train(x, n_dev, bn) for bn in bns]
        ys.append(ops.bn_relu_train(x, n_dev, bns[0]))
        ops.bn2d_relu_train(torch.randn(2, 8, 4, 4, device=dev), bn2)
        assert all(int(bn.num_batches_tracked) == 0 for bn in bns + [bn2])
    assert [int(bn.num_batches_tracked) for bn in bns + [bn2]] == [2, 1, 1, 1]
    ref = torch.nn.BatchNorm1d(16, eps=1e-3, momentum=0.01).to(dev).train()
    assert torch.allclose(ys[1], torch.relu(ref(x)), atol=2e-5)
    cma = torch.nn.BatchNorm1d(16, eps=1e-3, momentum=None).to(dev).train()   # cumulative average: counted at once
    with ops.deferred_batch_counts():
        ops.bn_relu_train(x, n_dev, cma)
        assert int(cma.num_batches_tracked) == 1


def test_sparse_sequential_with_the_fused_pair(dev):
    """spconv.SparseSequential.FUSED_BN_TRAIN: subm conv -> BatchNorm1d(train) -> ReLU -> strided conv, fused vs torch modules:
    same output features, same weight / BN gradients, same running statistics."""
    import numpy as np
    import spconv
    rng = np.random.RandomState(9)
    B, shape, n = 2, [11, 40, 36], 1200
    lin = rng.permutation(B * shape[0] * shape[1] * shape[2])[:n]
    idx = np.stack([lin // (shape[0] * shape[1] * shape[2]), (lin // (shape[1] * shape[2])) % shape[0], (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(n, 16, generator=g)
    w1, w2 = torch.randn(3, 3, 3, 16, 32, generator=g) * 0.1, torch.randn(3, 3, 3, 32, 64, generator=g) * 0.1
    results = []
    before = spconv.SparseSequential.FUSED_BN_TRAIN
    try:
        for fused in (False, True):
            spconv.SparseSequential.FUSED_BN_TRAIN = fused
            net = spconv.SparseSequential(spconv.SubMConv3d(16, 32, 3, bias=False, indice_key="s"), torch.nn.BatchNorm1d(32, eps=1e-3, momentum=0.01),
                                          torch.nn.ReLU(), spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False)).to(dev).train()
            with torch.no_grad():
                net[0].weight.copy_(w1.to(dev)); net[3].weight.copy_(w2.to(dev))
            out = net(spconv.SparseConvTensor(feat.to(dev), torch.from_numpy(idx).to(dev), shape, B))
            out.features.pow(2).sum().backward()
            order = np.lexsort(out.indices.cpu().numpy().T[::-1])
            results.append((out.features.detach().cpu()[order], net[0].weight.grad.cpu(), net[3].weight.grad.cpu(), net[1].weight.grad.cpu(),
                            net[1].bias.grad.cpu(), net[1].running_mean.cpu(), net[1].running_var.cpu()))
    finally:
        spconv.SparseSequential.FUSED_BN_TRAIN = before  # (round 2 left it False: the rest of a full run then used the torch modules)
    for a, b in zip(*results):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(a.abs().max()))), (a - b).abs().max()


@pytest.mark.parametrize("B,C,H,W", [(4, 128, 200, 176), (1, 256, 100, 88), (2, 1, 16, 12), (3, 24, 6, 10)])
@pytest.mark.parametrize("relu", [True, False])
def test_dense_layout_matches_torch_batchnorm2d(dev, B, C, H, W, relu):
    """sessd_bn2d_relu_train_fwd / _bwd vs torch.nn.BatchNorm2d(eps=1e-3, momentum=0.01) (+ ReLU) in train mode: output, running
    statistics, input / weight / bias gradients; bit-identical on a second run (fixed reduction order)."""
    g = torch.Generator().manual_seed(B * 7 + C)
    x = (torch.randn(B, C, H, W, generator=g) * 1.3 + torch.randn(1, C, 1, 1, generator=g)).to(dev)
    up = torch.randn(B, C, H, W, generator=g).to(dev)
    ref = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(dev).train()
    ref.weight.data.copy_(torch.rand(C, generator=g) + 0.5); ref.bias.data.copy_(torch.randn(C, generator=g) * 0.2)
    ref.running_mean.data.copy_(torch.randn(C, generator=g) * 0.1); ref.running_var.data.copy_(torch.rand(C, generator=g) + 0.5)
    mine = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(dev).train()
    mine.load_state_dict(ref.state_dict())
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    (yr * up).sum().backward()
    xm = x.clone().requires_grad_(True)
    ym = ops.bn2d_relu_train(xm, mine, relu)
    (ym * up).sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(ym, yr, rtol=0, atol=2e-5 * max(1.0, float(yr.abs().max())))
    assert torch.allclose(mine.running_mean, ref.running_mean, rtol=0, atol=1e-6)
    assert torch.allclose(mine.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    assert torch.allclose(xm.grad, xr.grad, rtol=0, atol=5e-5 * max(1.0, float(xr.grad.abs().max())))
    assert torch.allclose(mine.weight.grad, ref.weight.grad, rtol=2e-4, atol=2e-2 if B * H * W > 10000 else 1e-3)
    assert torch.allclose(mine.bias.grad, ref.bias.grad, rtol=2e-4, atol=2e-2 if B * H * W > 10000 else 1e-3)
    mine2 = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(dev).train()
    mine2.load_state_dict({k: v for k, v in ref.state_dict().items()})
    mine2.weight.data.copy_(mine.weight.data); mine2.bias.data.copy_(mine.bias.data)
    assert torch.equal(ops.bn2d_relu_train(x, mine2, relu), ym.detach())
    # a plane that is not a multiple of four goes through the torch module
    odd = torch.randn(2, C, 3, 5, generator=g).to(dev)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(dev).train()
    want = torch.relu(torch.nn.functional.batch_norm(odd, None, None, bn.weight, bn.bias, True, 0.01, 1e-3))
    assert torch.allclose(ops.bn2d_relu_train(odd, bn, True), want, atol=1e-5)


def test_ssfa_train_mode_fused_vs_torch_modules(dev):
    """SSFA in train mode with the fused BatchNorm2d + ReLU passes vs the torch modules: same output, same parameter gradients,
    same running statistics (the convolutions are the HIP kernels in both)."""
    import copy
    from det3d.models.necks.rpn_v1 import SSFA
    torch.manual_seed(3)
    a = SSFA([5], [1], [128], [1], [128], 128).to(dev).train()
    b = copy.deepcopy(a)
    a.fused_bn_train, b.fused_bn_train = True, False
    x = torch.randn(2, 128, 40, 48, device=dev)
    outs = []
    for m in (a, b):
        y = m(x)
        y = y[0] if isinstance(y, (tuple, list)) else y
        (y * torch.linspace(0.5, 1.5, y.numel(), device=dev).view_as(y)).sum().backward()
        outs.append(y.detach())
    assert torch.allclose(outs[0], outs[1], rtol=0, atol=2e-4 * max(1.0, float(outs[1].abs().max())))
    checked = 0
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and pa.grad is not None and pb.grad is not None
        # 13 BatchNorm layers deep, and a ReLU whose input differs in the last bit near zero switches a whole gradient path: single
        # entries may move by a few percent, so the two gradients are compared in norm
        err, ref = float((pa.grad - pb.grad).norm()), float(pb.grad.norm())
        assert torch.isfinite(pa.grad).all() and err <= 1e-2 * max(1e-3, ref), (na, err, ref)
        checked += 1
    for (na, ba), (nb, bb) in zip(a.named_buffers(), b.named_buffers()):
        if "running" in na:
            assert torch.allclose(ba, bb, rtol=1e-4, atol=1e-5), na
    assert checked > 30


@pytest.mark.parametrize("B,C,H,W", [(4, 128, 200, 176), (1, 128, 200, 176), (2, 8, 6, 10), (3, 64, 4, 5)])
def test_ssfa_attention_tail_train_mode(dev, B, C, H, W):
    """ops.ssfa_fuse_train (csrc/ssfa_train.hip) vs the torch composition of rpn_v1.py:225-235 in train mode: output, the running
    statistics of both BatchNorm2d(1), gradients of both inputs, both conv weights and the four BatchNorm parameters; a second
    run gives the same bits."""
    nn = torch.nn
    g = torch.Generator().manual_seed(B * 100 + C)
    x0 = (torch.randn(B, C, H, W, generator=g) * 1.2 + 0.2).to(dev)
    x1 = (torch.randn(B, C, H, W, generator=g) * 0.8 - 0.1).to(dev)
    up = torch.randn(B, C, H, W, generator=g).to(dev)

    def branches():
        torch.manual_seed(5)
        mods = []
        for k in range(2):
            conv, bn = nn.Conv2d(C, 1, 1, bias=False), nn.BatchNorm2d(1, eps=1e-3, momentum=0.01)
            bn.weight.data.fill_(1.3 - 0.5 * k); bn.bias.data.fill_(0.2 * k - 0.1)
            bn.running_mean.data.fill_(0.05 * (k + 1)); bn.running_var.data.fill_(0.7 + 0.2 * k)
            mods += [conv.to(dev), bn.to(dev).train()]
        return mods

    ref, mine = branches(), branches()
    a0, a1 = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
    w = torch.softmax(torch.cat([ref[1](ref[0](a0)), ref[3](ref[2](a1))], dim=1), dim=1)
    yr = a0 * w[:, 0:1] + a1 * w[:, 1:]
    (yr * up).sum().backward()
    m0, m1 = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
    assert ops.ssfa_fuse_train_covers(m0, *mine)
    ym = ops.ssfa_fuse_train(m0, m1, *mine)
    (ym * up).sum().backward()
    torch.cuda.synchronize()
    tol = lambda t, r: r * max(1.0, float(t.abs().max()))
    assert torch.allclose(ym, yr, rtol=0, atol=tol(yr, 2e-5))
    for k in (1, 3):
        assert torch.allclose(mine[k].running_mean, ref[k].running_mean, rtol=0, atol=1e-6)
        assert torch.allclose(mine[k].running_var, ref[k].running_var, rtol=1e-5, atol=1e-6)
        assert int(mine[k].num_batches_tracked) == int(ref[k].num_batches_tracked) == 1
    assert torch.allclose(m0.grad, a0.grad, rtol=0, atol=tol(a0.grad, 5e-5))
    assert torch.allclose(m1.grad, a1.grad, rtol=0, atol=tol(a1.grad, 5e-5))
    for pm, pr in zip([p for m in mine for p in m.parameters()], [p for m in ref for p in m.parameters()]):
        assert pm.grad is not None and pm.grad.shape == pr.grad.shape
        assert torch.allclose(pm.grad, pr.grad, rtol=2e-3, atol=tol(pr.grad, 2e-3)), (pm.grad.flatten()[:4], pr.grad.flatten()[:4])
    again = branches()
    y2 = ops.ssfa_fuse_train(x0, x1, *again)
    assert torch.equal(y2, ym.detach())
    # not covered: a biased conv, an eval-mode BatchNorm
    assert not ops.ssfa_fuse_train_covers(x0, nn.Conv2d(C, 1, 1).to(dev), mine[1], mine[2], mine[3])
    assert not ops.ssfa_fuse_train_covers(x0, mine[0], mine[1].eval(), mine[2], mine[3])


def test_dense_backward_mask_from_x_equals_mask_from_y(dev):
    """sessd_bn2d_relu_train_bwd_x (ReLU mask re-derived from x) == sessd_bn2d_relu_train_bwd (mask read from the forward output):
    the same bits for dx, dgamma, dbeta -- inputs with many values at and around the ReLU threshold."""
    from sessd_hip._lib import lib, check
    B, C, H, W = 2, 16, 24, 40
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(B, C, H, W, generator=g) * 1e-3).to(dev)          # z close to beta: both signs, tiny magnitudes
    x[:, ::2] += torch.randn(B, C // 2, H, W, generator=g).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 1e-4).to(dev)
    dy = torch.randn(B, C, H, W, generator=g).to(dev)
    y, mean, invstd = torch.empty_like(x), torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = ops.zeroed_workspace(lib.sessd_bn2d_relu_train_workspace_bytes(C), dev, "bn2d")
    st = torch.cuda.current_stream().cuda_stream
    check(lib.sessd_bn2d_relu_train_fwd(x.data_ptr(), B, C, H * W, gamma.data_ptr(), beta.data_ptr(), 1e-3, 0.01, 1, None, None,
                                        y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), st), "fwd")
    outs = []
    for from_x in (False, True):
        dx, dg, db = torch.empty_like(x), torch.empty(C, device=dev), torch.empty(C, device=dev)
        if from_x:
            check(lib.sessd_bn2d_relu_train_bwd_x(dy.data_ptr(), x.data_ptr(), B, C, H * W, gamma.data_ptr(), beta.data_ptr(),
                                                  mean.data_ptr(), invstd.data_ptr(), 1, dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), st), "bwd_x")
        else:
            check(lib.sessd_bn2d_relu_train_bwd(dy.data_ptr(), x.data_ptr(), y.data_ptr(), B, C, H * W, gamma.data_ptr(), mean.data_ptr(),
                                                invstd.data_ptr(), 1, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                ws.numel(), st), "bwd")
        outs.append((dx, dg, db))
    torch.cuda.synchronize()
    assert 0.2 < float((y > 0).float().mean()) < 0.8
    for a, b in zip(*outs):
        assert torch.equal(a, b)
