"""Backward of the dense BEV convolutions on the HIP kernels (ops.Conv2dFunction: data gradient through the forward
kernels with re-packed weights, weight gradient by sessd_conv2d_wgrad) vs torch autograd on CPU (F.conv2d /
F.conv_transpose2d: the ops the reference's SSFA calls, rpn_v1.py:135-235). Tolerance 2e-4 of the largest magnitude
(float32 sums over up to 35200 pixels x batch in another order; Winograd rounding on the eligible 3x3 layers)."""
import pytest
import torch
import torch.nn.functional as F

from sessd_hip import ops

pytestmark = pytest.mark.gpu


def _close(got, want, what):
    scale = max(1e-6, float(want.abs().max()))
    err = float((got.cpu() - want).abs().max())
    assert err <= 2e-4 * scale, (what, err, scale)


CASES = [
    # kind, cin, cout, k, stride, H, W, bias
    ("conv", 16, 32, 3, 1, 24, 40, False),
    ("conv", 128, 128, 3, 1, 64, 64, False),   # Winograd-eligible size (H*W >= 4096, cin % 8 == 0)
    ("conv", 16, 48, 3, 2, 32, 48, False),
    ("conv", 32, 16, 1, 1, 24, 40, False),
    ("conv", 16, 22, 1, 1, 16, 24, True),
    ("conv", 128, 128, 1, 1, 40, 48, False),   # 1x1 with >= 64 channels: the 64 x 64 wave-tile weight-gradient kernel
    ("conv", 64, 160, 1, 1, 12, 24, True),    # ... with a ragged channel block
    ("deconv", 32, 16, 3, 2, 12, 24, False),
    ("deconv", 256, 128, 3, 2, 20, 16, False),   # cin, cout % 64 == 0: the LDS-staged stride-2 weight-gradient kernel
    ("conv", 128, 256, 3, 2, 40, 48, False),     # ... called directly (b1.0 of the neck), 7 row chunks of ragged length
    ("conv", 64, 64, 3, 2, 16, 16, True),
]


@pytest.mark.parametrize("kind,cin,cout,k,stride,H,W,bias", CASES)
@pytest.mark.parametrize("batch", [1, 3])
def test_conv_gradients(dev, kind, cin, cout, k, stride, H, W, bias, batch):
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(batch, cin, H, W, generator=g)
    if kind == "conv":
        m = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=bias)
    else:
        m = torch.nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=bias)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.1)
        if bias:
            m.bias.copy_(torch.randn(cout, generator=g))
    xr = x.clone().requires_grad_(True)
    yr = m(xr)
    gy = torch.randn(yr.shape, generator=g)
    (yr * gy).sum().backward()
    md = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=bias) if kind == "conv" else \
        torch.nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=bias)
    md.load_state_dict(m.state_dict())
    md = md.to(dev)
    xd = x.to(dev).requires_grad_(True)
    yd = ops.conv2d_module(xd, md)
    _close(yd.detach(), yr.detach(), "forward")
    (yd * gy.to(dev)).sum().backward()
    _close(md.weight.grad, m.weight.grad, "weight grad")
    _close(xd.grad, xr.grad, "input grad")
    if bias:
        _close(md.bias.grad, m.bias.grad, "bias grad")


def test_wgrad_is_deterministic(dev):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 40, 48, generator=g).to(dev)
    gy = torch.randn(2, 64, 40, 48, generator=g).to(dev)
    a = ops.conv2d_wgrad(x, gy, 3, 1)
    b = ops.conv2d_wgrad(x, gy, 3, 1)
    assert torch.equal(a, b)
    want = torch.nn.grad.conv2d_weight(x.cpu(), (64, 64, 3, 3), gy.cpu(), stride=1, padding=1)
    _close(a, want, "wgrad")


def test_train_mode_heads_fused_equal_the_four_torch_convs(dev):
    """det3d Head in train mode: the four 1x1 convs as ONE 22-channel conv through ops.Conv2dFunction (forward, data gradient,
    weight gradient on the HIP kernels, bias gradient on sessd_nchw_channel_sum) against the four torch modules (MIOpen):
    the four NHWC outputs, the gradients of all eight parameter tensors and of the input."""
    from det3d.models.bbox_heads.mg_head_sessd import Head
    torch.manual_seed(3)
    head = Head(128, 14, 2, use_dir=True, num_dir=4).to(dev).train()
    x = torch.randn(2, 128, 200, 176, device=dev)
    gos = None
    res = []
    for fused in (False, True):
        head.fused_train = fused
        for p in head.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        out = head(xi)
        assert [tuple(out[k].shape) for k in ("box_preds", "cls_preds", "dir_cls_preds", "iou_preds")] == \
            [(2, 200, 176, 14), (2, 200, 176, 2), (2, 200, 176, 4), (2, 200, 176, 2)]
        if gos is None:
            gos = {k: torch.randn_like(v) for k, v in out.items()}
        sum((out[k] * gos[k]).sum() for k in out).backward()
        res.append(({k: v.detach().clone() for k, v in out.items()}, {n: p.grad.clone() for n, p in head.named_parameters()}, xi.grad.clone()))
    (o0, g0, x0), (o1, g1, x1) = res
    for k in o0:
        _close(o1[k], o0[k].cpu(), k)
    for n in g0:
        _close(g1[n], g0[n].cpu(), n)
    _close(x1, x0.cpu(), "input grad")


def test_channel_sum(dev):
    x = torch.randn(3, 22, 200, 176, device=dev)
    got = ops.channel_sum(x)
    want = x.double().sum((0, 2, 3))
    assert float((got.double() - want).abs().max()) <= 1e-6 * float(x.double().abs().sum((0, 2, 3)).max())
    odd = torch.randn(2, 5, 7, 9, device=dev)   # H * W not a multiple of 4: torch's reduction
    assert torch.allclose(ops.channel_sum(odd), odd.sum((0, 2, 3)), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,cin,cout,H,W", [(4, 128, 128, 200, 176), (2, 256, 256, 100, 88), (1, 64, 128, 8, 16), (3, 128, 64, 10, 24),
                                             (1, 64, 64, 2, 16), (2, 64, 64, 6, 20), (5, 64, 64, 4, 18)])
def test_winograd_domain_weight_gradient(dev, B, cin, cout, H, W):
    """sessd_conv3x3_wgrad_winograd (dU = sum over tiles of (A dY A^T)(B^T d B), then G^T dU G) vs the direct pixel-reduction
    kernel sessd_conv2d_wgrad and, on the small shapes, vs torch autograd on the CPU: 2e-4 of the largest entry; the image
    borders (tiles whose patch leaves the image), stages that straddle rows / images and ragged last stages are in the shapes.
    A second run gives the same bits."""
    g = torch.Generator().manual_seed(B * 1000 + cin + H)
    x = torch.randn(B, cin, H, W, generator=g)
    gy = torch.randn(B, cout, H, W, generator=g)
    x[:, :, 0, :] += 2.0; x[:, :, -1, :] -= 1.5; x[:, :, :, 0] += 1.0; x[:, :, :, -1] -= 2.5   # borders must count
    xd, gd = x.to(dev), gy.to(dev)
    wino = ops.conv2d_wgrad(xd, gd, 3, 1, winograd=True)
    if W % 8 == 0:   # the direct kernel's own constraint
        _close(wino, ops.conv2d_wgrad(xd, gd, 3, 1, winograd=False).cpu(), "winograd-domain vs direct")
    if B * H * W <= 4096:
        w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
        (F.conv2d(x, w, padding=1) * gy).sum().backward()
        _close(wino, w.grad, "winograd-domain vs torch")
    assert torch.equal(ops.conv2d_wgrad(xd, gd, 3, 1, winograd=True), wino)
    with pytest.raises(ValueError):
        ops.conv2d_wgrad(xd[:, :48].contiguous(), gd, 3, 1, winograd=True)


def test_winograd_forward_at_the_last_pixel(dev):
    """The 16-byte patch loads of the Winograd forward kernels at the right border of the LAST row of the LAST channel end
    exactly at the tensor's end: a value there must reach the output (first-generation and stream-K kernels vs the direct one)."""
    B, C, H, W = 1, 128, 64, 64
    x = torch.zeros(B, C, H, W)
    x[0, C - 1, H - 1, W - 1] = 3.0
    x[0, C - 1, H - 1, W - 2] = -2.0
    x[0, 0, 0, 0] = 1.5
    w = torch.randn(C, C, 3, 3, generator=torch.Generator().manual_seed(1)) * 0.1
    want = F.conv2d(x, w, padding=1)
    pc = ops.pack_conv2d(w.to(dev), 1)
    for cfg in (None, 20, 22):
        got = ops.conv2d(x.to(dev), pc, None, None, False, tile_cfg=cfg)
        _close(got, want, "tile_cfg %s" % cfg)


def test_split_nhwc_and_its_adjoint(dev):
    """ops.split_nhwc (sessd_nchw_split_nhwc / sessd_nhwc_merge_nchw) == [p.permute(0, 2, 3, 1).contiguous() for p in split]: the
    same bits forward, the same gradient, a part without a gradient contributes zeros."""
    g = torch.Generator().manual_seed(2)
    y = torch.randn(3, 22, 10, 12, generator=g).to(dev)
    sizes = [14, 2, 4, 2]
    a = y.clone().requires_grad_(True)
    b = y.clone().requires_grad_(True)
    pa = ops.split_nhwc(a, sizes)
    pb = [p.permute(0, 2, 3, 1).contiguous() for p in torch.split(b, sizes, dim=1)]
    assert all(u.is_contiguous() and torch.equal(u, v) for u, v in zip(pa, pb))
    ws = [torch.randn(p.shape, generator=g).to(dev) for p in pb]
    (pa[0] * ws[0]).sum().add((pa[2] * ws[2]).sum()).add((pa[3] * ws[3]).sum()).backward()     # part 1 unused
    (pb[0] * ws[0]).sum().add((pb[2] * ws[2]).sum()).add((pb[3] * ws[3]).sum()).backward()
    assert torch.equal(a.grad, b.grad) and float(a.grad[:, 14:16].abs().sum()) == 0
    one = ops.split_nhwc(y, [22])
    assert torch.equal(one[0], y.permute(0, 2, 3, 1).contiguous())
