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
    ("deconv", 32, 16, 3, 2, 12, 24, False),
    ("deconv", 256, 128, 3, 2, 20, 16, False),
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
