"""HIP f32-MFMA BEV convolutions vs torch CPU float32 (the same ops the reference's SSFA / Head call).
Tolerance: float32 sums of K <= 2304 products in a different order: 2e-4 * max|ref| absolute."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from sessd_hip import ops

pytestmark = pytest.mark.gpu


def _close(got, ref, rel=2e-4):
    tol = rel * max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err < tol, (err, tol)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("cin,cout,k,stride,H,W", [(128, 128, 3, 1, 24, 40), (128, 256, 3, 2, 24, 40), (256, 256, 3, 1, 12, 20),
                                                   (128, 128, 1, 1, 24, 40), (6, 32, 3, 1, 7, 9)])
def test_conv2d(dev, cfg, cin, cout, k, stride, H, W):
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.05
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(F.conv2d(x, w, stride=stride, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    pc = ops.pack_conv2d(w.to(dev), stride)
    got = ops.conv2d(x.to(dev), pc, scale.to(dev), shift.to(dev), True, tile_cfg=cfg).cpu()
    assert got.shape == ref.shape
    _close(got, ref)


@pytest.mark.parametrize("cin,cout,H,W", [(128, 128, 24, 40), (256, 256, 13, 19), (16, 32, 9, 7), (128, 128, 200, 176)])
def test_lds_variant_bit_identical(dev, cin, cout, H, W):
    """tile_cfg 10 (activation-stationary, LDS staged) must equal the direct kernel bit for bit (same fmaf order)."""
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(2, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
    shift = (torch.randn(cout, generator=g) * 0.1).to(dev)
    res = torch.randn(2, cout, H, W, generator=g).to(dev)
    pc = ops.pack_conv2d(w, 1)
    a = ops.conv2d(x, pc, scale, shift, True, residual=res, tile_cfg=3)
    b = ops.conv2d(x, pc, scale, shift, True, residual=res, tile_cfg=10)
    assert torch.equal(a, b)
    ref = torch.relu(F.conv2d(x.cpu(), w.cpu(), padding=1) * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1)) + res.cpu()
    _close(b.cpu(), ref)


@pytest.mark.parametrize("cin,cout,H,W,B", [(128, 128, 24, 40, 2), (256, 256, 14, 18, 1), (8, 32, 6, 4, 1), (128, 128, 200, 176, 1), (16, 40, 10, 66, 3)])
def test_winograd_variant(dev, cin, cout, H, W, B):
    """tile_cfg 20: fused Winograd F(2x2,3x3). Not bit-identical to the direct kernel (different rounding): compared with
    float64 torch conv at 3e-6 * max|ref| * sqrt(K)/8 -- and it must not be worse than 4x the direct kernel's own error."""
    g = torch.Generator().manual_seed(cin + H + W)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g)
    ref = (torch.relu(F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
           + res.double())
    pc = ops.pack_conv2d(w.to(dev), 1)
    args = (x.to(dev), pc, scale.to(dev), shift.to(dev), True)
    wino = ops.conv2d(*args, residual=res.to(dev), tile_cfg=20).cpu().double()
    direct = ops.conv2d(*args, residual=res.to(dev), tile_cfg=3).cpu().double()
    e_w, e_d = float((wino - ref).abs().max()), float((direct - ref).abs().max())
    print("winograd err %.2e  direct err %.2e  (max|ref| %.2f)" % (e_w, e_d, float(ref.abs().max())))
    assert e_w < 2e-4 * max(1.0, float(ref.abs().max()))
    assert e_w < 8 * e_d + 1e-6


@pytest.mark.parametrize("cin,cout,H,W,B,wgs", [(128, 128, 24, 40, 2, 0), (256, 256, 14, 18, 1, 0), (16, 40, 10, 66, 3, 8), (128, 128, 200, 176, 1, 0),
                                                 (128, 128, 200, 176, 1, 104), (64, 300, 12, 16, 1, 64), (32, 128, 8, 8, 1, 16), (256, 256, 100, 88, 2, 0)])
@pytest.mark.parametrize("cfg", [22, 23, 24])
def test_winograd_stream_k(dev, cin, cout, H, W, B, wgs, cfg):
    """tile_cfg 22 / 23: 128 / 64 couts per workgroup of 8 / 4 waves, rounds dealt out in equal shares (stream-K); tile_cfg 24: the
    third generation (4 waves, a wave owns all 16 transform points of its 32 couts: output transform in registers, no LDS exchange
    in the epilogue) -- same units, rounds and cuts as 22, so its results must equal 22's BIT FOR BIT. Same bound as the first-generation
    Winograd kernel; covers whole units, units cut in two and (few rounds per share) units cut in three or more parts; the
    second and third launch reuse the workspace (the counters must be back at zero) and must give the same bits."""
    g = torch.Generator().manual_seed(cin + H + W + wgs)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g)
    ref = (torch.relu(F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
           + res.double())
    pc = ops.pack_conv2d(w.to(dev), 1)
    args = (x.to(dev), pc, scale.to(dev), shift.to(dev), True)
    ws = ops.winograd_sk_workspace(B, H, W, cout, dev, wgs, cfg - 22)
    a = ops.conv2d(*args, residual=res.to(dev), tile_cfg=cfg, workspace=ws, workgroups=wgs)
    b = ops.conv2d(*args, residual=res.to(dev), tile_cfg=cfg, workspace=ws, workgroups=wgs)
    c = ops.conv2d(*args, residual=res.to(dev), tile_cfg=cfg, workspace=ws, workgroups=wgs)
    torch.cuda.synchronize()
    units = B * (((H // 2) * (W // 2) + 31) // 32) * ((cout + 127) // 128 if cfg != 23 else (cout + 63) // 64)
    assert int(ws[:units * 4].view(torch.int32).abs().sum().item()) == 0       # counters left at zero
    assert torch.equal(a, b) and torch.equal(a, c)
    if cfg == 24:
        ws22 = ops.winograd_sk_workspace(B, H, W, cout, dev, wgs, 0)
        a22 = ops.conv2d(*args, residual=res.to(dev), tile_cfg=22, workspace=ws22, workgroups=wgs)
        assert torch.equal(a, a22), "tile_cfg 24 must reproduce tile_cfg 22's bits (max diff %.3e)" % float((a - a22).abs().max())
        p24 = ops.conv2d(x.to(dev), pc, None, None, False, tile_cfg=24, workspace=ws, workgroups=wgs)
        p22 = ops.conv2d(x.to(dev), pc, None, None, False, tile_cfg=22, workspace=ws22, workgroups=wgs)
        assert torch.equal(p24, p22)
    direct = ops.conv2d(*args, residual=res.to(dev), tile_cfg=3).cpu().double()
    e_w, e_d = float((a.cpu().double() - ref).abs().max()), float((direct - ref).abs().max())
    print("stream-K winograd err %.2e  direct err %.2e  (max|ref| %.2f)" % (e_w, e_d, float(ref.abs().max())))
    assert e_w < 2e-4 * max(1.0, float(ref.abs().max()))
    assert e_w < 8 * e_d + 1e-6
    # no ReLU / BatchNorm / residual
    plain = ops.conv2d(x.to(dev), pc, None, None, False, tile_cfg=cfg, workspace=ws, workgroups=wgs).cpu().double()
    assert float((plain - F.conv2d(x.double(), w.double(), padding=1)).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape,B,H,W,wgs", [(0, 1, 200, 176, 0), (1, 2, 24, 40, 0), (0, 2, 10, 66, 8), (1, 1, 14, 18, 16), (2, 1, 200, 176, 0),
                                             (2, 2, 10, 66, 8)])
def test_winograd_stream_k_weight_sets(dev, shape, B, H, W, wgs):
    """sessd_conv3x3_winograd_sk_sets: two layers of one shape (conv_0 / conv_1 of the SSFA neck) as ONE stream-K launch over a
    (2 B, C, H, W) input, each half with its own weights and BatchNorm constants, against the two single-layer launches."""
    C = 128
    g = torch.Generator().manual_seed(shape + H)
    x = torch.randn(2 * B, C, H, W, generator=g).to(dev)
    ws_ = [(torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev) for _ in range(2)]
    sc = torch.stack([torch.rand(C, generator=g) + 0.5 for _ in range(2)]).to(dev)
    sh = torch.stack([torch.randn(C, generator=g) * 0.1 for _ in range(2)]).to(dev)
    pcs = [ops.pack_conv2d(w_, 1) for w_ in ws_]
    upk = torch.cat([pc.upk_sk(shape).reshape(-1) for pc in pcs])
    ws = ops.winograd_sk_workspace(2 * B, H, W, C, dev, wgs, shape)
    out = torch.empty_like(x)
    ops.conv2d_winograd_sk_sets(x, upk, 2, C, sc, sh, True, out, shape, ws, wgs)
    again = torch.empty_like(x)
    ops.conv2d_winograd_sk_sets(x, upk, 2, C, sc, sh, True, again, shape, ws, wgs)
    assert torch.equal(out, again)
    for s_ in range(2):
        one = ops.conv2d(x[s_ * B:(s_ + 1) * B].contiguous(), pcs[s_], sc[s_].contiguous(), sh[s_].contiguous(), True, tile_cfg=22 + shape,
                         workspace=ws, workgroups=wgs)
        ref = torch.relu(F.conv2d(x[s_ * B:(s_ + 1) * B].cpu().double(), ws_[s_].cpu().double(), padding=1) * sc[s_].cpu().double().view(1, -1, 1, 1)
                         + sh[s_].cpu().double().view(1, -1, 1, 1))
        e_m = float((out[s_ * B:(s_ + 1) * B].cpu().double() - ref).abs().max())
        e_1 = float((one.cpu().double() - ref).abs().max())
        assert e_m < 2e-4 * max(1.0, float(ref.abs().max())) and e_m < 4 * e_1 + 1e-6


@pytest.mark.parametrize("cin,cout,k,stride,H,W,B,wgs", [(128, 256, 3, 2, 24, 40, 2, 0), (128, 128, 1, 1, 24, 40, 1, 0), (256, 256, 1, 1, 13, 19, 2, 8),
                                                         (16, 40, 3, 1, 10, 66, 3, 8), (32, 130, 3, 2, 9, 7, 1, 16), (128, 256, 3, 2, 200, 176, 1, 0),
                                                         (64, 22, 1, 1, 30, 20, 1, 64), (128, 128, 3, 1, 24, 40, 1, 248)])
def test_conv2d_lds_stream_k(dev, cin, cout, k, stride, H, W, B, wgs):
    """tile_cfg 30 (csrc/dense_conv_sk.hip): LDS-tiled implicit GEMM, rounds dealt out in equal shares. Whole units, units cut in
    two and (few rounds per share) in three or more parts, partial pixel tiles and cout groups; the workspace is reused by three
    launches (counters back at zero) that must give the same bits."""
    g = torch.Generator().manual_seed(cin + H + W + wgs)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.05
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    conv = F.conv2d(x.double(), w.double(), stride=stride, padding=k // 2)
    res = torch.randn(conv.shape, generator=g)
    ref = torch.relu(conv * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)) + res.double()
    pc = ops.pack_conv2d(w.to(dev), stride)
    Ho, Wo = conv.shape[2], conv.shape[3]
    ws = ops.conv2d_sk_workspace(B, Ho, Wo, cout, 1, dev, wgs)
    args = (x.to(dev), pc, scale.to(dev), shift.to(dev), True)
    a = ops.conv2d(*args, residual=res.to(dev), tile_cfg=30, workspace=ws, workgroups=wgs)
    b = ops.conv2d(*args, residual=res.to(dev), tile_cfg=30, workspace=ws, workgroups=wgs)
    c = ops.conv2d(*args, residual=res.to(dev), tile_cfg=30, workspace=ws, workgroups=wgs)
    torch.cuda.synchronize()
    units = B * ((Ho * Wo + 127) // 128) * ((cout + 127) // 128)
    assert int(ws[:units * 4].view(torch.int32).abs().sum().item()) == 0       # counters left at zero
    assert torch.equal(a, b) and torch.equal(a, c)
    direct = ops.conv2d(*args, residual=res.to(dev), tile_cfg=3).cpu().double()
    e_s, e_d = float((a.cpu().double() - ref).abs().max()), float((direct - ref).abs().max())
    print("LDS stream-K err %.2e  direct err %.2e  (max|ref| %.2f)" % (e_s, e_d, float(ref.abs().max())))
    assert e_s < 4 * e_d + 1e-6          # exact float32 arithmetic in a different summation order
    plain = ops.conv2d(x.to(dev), pc, None, None, False, tile_cfg=30, workspace=ws, workgroups=wgs).cpu().double()
    assert float((plain - conv).abs().max()) < 4 * e_d + 1e-6


@pytest.mark.parametrize("B,H,W,wgs", [(2, 10, 12, 0), (1, 100, 88, 0), (1, 7, 9, 8), (3, 16, 8, 40)])
def test_deconv_s2_lds_stream_k(dev, B, H, W, wgs):
    """The four output-parity classes of ConvTranspose2d(256, 128, 3, 2, 1, 1) as one stream-K launch (tile_cfg 30)."""
    g = torch.Generator().manual_seed(9 + H)
    x = torch.randn(B, 256, H, W, generator=g)
    w = torch.randn(256, 128, 3, 3, generator=g) * 0.05
    scale = torch.rand(128, generator=g) + 0.5
    shift = torch.randn(128, generator=g) * 0.1
    res = torch.randn(B, 128, 2 * H, 2 * W, generator=g)
    ref = torch.relu(F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1, output_padding=1) * scale.double().view(1, -1, 1, 1)
                     + shift.double().view(1, -1, 1, 1)) + res.double()
    pc = ops.pack_deconv2d_s2(w.to(dev))
    ws = ops.conv2d_sk_workspace(B, H, W, 128, 4, dev, wgs)
    args = (x.to(dev), pc, scale.to(dev), shift.to(dev), True)
    a = ops.conv2d(*args, residual=res.to(dev), tile_cfg=30, workspace=ws, workgroups=wgs)
    b = ops.conv2d(*args, residual=res.to(dev), tile_cfg=30, workspace=ws, workgroups=wgs)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    direct = ops.conv2d(*args, residual=res.to(dev), tile_cfg=4).cpu().double()
    e_s, e_d = float((a.cpu().double() - ref).abs().max()), float((direct - ref).abs().max())
    print("LDS stream-K deconv err %.2e  direct err %.2e" % (e_s, e_d))
    assert e_s < 4 * e_d + 1e-6


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 10, 12, 256, 128), (1, 100, 88, 256, 128), (1, 7, 9, 16, 40), (3, 16, 8, 64, 22)])
@pytest.mark.parametrize("cfg", [40, 41, 42])
def test_deconv_s2_paired_classes_bit_identical(dev, cfg, B, H, W, cin, cout):
    """tile_cfg 40 .. 42 (csrc/dense_deconv_pair.hip; three workgroup shapes): both px classes of a row parity in every wave, 8-byte stores. Per class the
    same fmaf chain as the class-per-workgroup launch: equal bits, with and without BatchNorm / ReLU / residual."""
    g = torch.Generator().manual_seed(cfg + H)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cin, cout, 3, 3, generator=g) * 0.05).to(dev)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
    res = torch.randn(B, cout, 2 * H, 2 * W, generator=g).to(dev)
    pc = ops.pack_deconv2d_s2(w)
    a = ops.conv2d(x, pc, scale, shift, True, residual=res, tile_cfg=cfg)
    b = ops.conv2d(x, pc, scale, shift, True, residual=res, tile_cfg=4)
    assert torch.equal(a, b)
    a = ops.conv2d(x, pc, None, None, False, tile_cfg=cfg)
    b = ops.conv2d(x, pc, None, None, False, tile_cfg=4)
    assert torch.equal(a, b)
    ref = F.conv_transpose2d(x.cpu().double(), w.cpu().double(), stride=2, padding=1, output_padding=1)
    assert float((a.cpu().double() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cfg", [3, 4, 11, 12])
@pytest.mark.parametrize("B,H,W", [(2, 10, 12), (1, 100, 88)])
def test_deconv_s2_two_layers_one_launch(dev, cfg, B, H, W):
    """sessd_deconv2d_s2_mfma_pair: deconv_block_0 (+ residual) and deconv_block_1 of the SSFA neck read the same input; one launch
    over their 2 x 4 parity classes gives the bits of the two single-layer launches."""
    g = torch.Generator().manual_seed(cfg + H)
    x = torch.randn(B, 256, H, W, generator=g).to(dev)
    pcs = [ops.pack_deconv2d_s2((torch.randn(256, 128, 3, 3, generator=g) * 0.05).to(dev)) for _ in range(2)]
    sc = [(torch.rand(128, generator=g) + 0.5).to(dev) for _ in range(2)]
    sh = [(torch.randn(128, generator=g) * 0.1).to(dev) for _ in range(2)]
    res = torch.randn(B, 128, 2 * H, 2 * W, generator=g).to(dev)
    a = ops.conv2d(x, pcs[0], sc[0], sh[0], True, residual=res, tile_cfg=cfg)
    b = ops.conv2d(x, pcs[1], sc[1], sh[1], True, tile_cfg=cfg)
    oa, ob = torch.empty_like(a), torch.empty_like(b)
    ops.deconv2d_s2_pair(x, pcs[0], pcs[1], sc[0], sh[0], sc[1], sh[1], True, oa, ob, residual_a=res, tile_cfg=cfg)
    assert torch.equal(oa, a) and torch.equal(ob, b)


@pytest.mark.parametrize("cfg", [1, 3, 4, 11, 12])
def test_deconv_s2_with_residual(dev, cfg):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 256, 10, 12, generator=g)
    w = torch.randn(256, 128, 3, 3, generator=g) * 0.05
    scale = torch.rand(128, generator=g) + 0.5
    shift = torch.randn(128, generator=g) * 0.1
    res = torch.randn(2, 128, 20, 24, generator=g)
    ref = torch.relu(F.conv_transpose2d(x, w, stride=2, padding=1, output_padding=1) * scale.view(1, -1, 1, 1)
                     + shift.view(1, -1, 1, 1)) + res
    pc = ops.pack_deconv2d_s2(w.to(dev))
    got = ops.conv2d(x.to(dev), pc, scale.to(dev), shift.to(dev), True, residual=res.to(dev), tile_cfg=cfg).cpu()
    _close(got, ref)


def test_head_conv_cout22_bias_no_relu(dev):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 128, 16, 24, generator=g)
    w = torch.randn(22, 128, 1, 1, generator=g) * 0.1
    b = torch.randn(22, generator=g)
    ref = F.conv2d(x, w, b)
    got = ops.conv2d(x.to(dev), ops.pack_conv2d(w.to(dev)), None, b.to(dev), False).cpu()
    _close(got, ref)


def test_transpose_detecting_identity(dev):
    """A = I style check with an asymmetric operand: 1x1 conv with a permutation-like weight."""
    cin = cout = 64
    w = torch.zeros(cout, cin, 1, 1)
    for o in range(cout):
        w[o, (o * 7 + 3) % cin, 0, 0] = 1.0 + o
    x = torch.arange(cin * 8 * 8, dtype=torch.float32).view(1, cin, 8, 8) * 1e-3
    ref = F.conv2d(x, w)
    got = ops.conv2d(x.to(dev), ops.pack_conv2d(w.to(dev)), None, None, False).cpu()
    assert torch.equal(got, ref)


def test_ssfa_fuse(dev):
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(2, 128, 10, 12, generator=g)
    x1 = torch.randn(2, 128, 10, 12, generator=g)
    w0 = torch.randn(128, generator=g) * 0.1
    w1 = torch.randn(128, generator=g) * 0.1
    s0, t0, s1, t1 = 1.3, -0.2, 0.7, 0.1
    a0 = (x0 * w0.view(1, -1, 1, 1)).sum(1, keepdim=True) * s0 + t0
    a1 = (x1 * w1.view(1, -1, 1, 1)).sum(1, keepdim=True) * s1 + t1
    sm = torch.softmax(torch.cat([a0, a1], 1), 1)
    ref = x0 * sm[:, 0:1] + x1 * sm[:, 1:]
    got = ops.ssfa_fuse(x0.to(dev), x1.to(dev), w0.to(dev), w1.to(dev), s0, t0, s1, t1).cpu()
    _close(got, ref, 1e-5)


@pytest.mark.parametrize("B,C,H,W", [(1, 128, 200, 176), (2, 128, 9, 7), (3, 64, 5, 13)])
def test_ssfa_fuse_with_heads(dev, B, C, H, W):
    """sessd_ssfa_fuse_head: the SSFA fusion tail (rpn_v1.py:227-233) and the four 1x1 heads (mg_head_sessd.py:217-230) in one launch
    against the two-launch form and a float64 restatement; with and without the optional SSFA output."""
    g = torch.Generator().manual_seed(B + C)
    x0, x1 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    w0, w1 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    s0, t0, s1, t1 = 1.3, -0.2, 0.7, 0.1
    hw, hb = torch.randn(22, C, generator=g) * 0.05, torch.randn(22, generator=g) * 0.1
    a = (x0.double() * w0.double().view(1, -1, 1, 1)).sum(1) * s0 + t0
    b = (x1.double() * w1.double().view(1, -1, 1, 1)).sum(1) * s1 + t1
    p = torch.softmax(torch.stack([a, b], 1), 1)
    ssfa = x0.double() * p[:, 0:1] + x1.double() * p[:, 1:2]
    head = torch.einsum("oc,bchw->bohw", hw.double(), ssfa) + hb.double().view(1, -1, 1, 1)
    d = lambda t: t.to(dev)
    out = torch.full((B, C, H, W), float("nan"), device=dev)
    got = ops.ssfa_fuse_head(d(x0), d(x1), d(w0), d(w1), s0, t0, s1, t1, d(hw), d(hb), out=out)
    assert float((out.cpu().double() - ssfa).abs().max()) < 1e-5 * max(1.0, float(ssfa.abs().max()))
    assert float((got.cpu().double().view(B, 22, H, W) - head).abs().max()) < 2e-5 * max(1.0, float(head.abs().max()))
    again = ops.ssfa_fuse_head(d(x0), d(x1), d(w0), d(w1), s0, t0, s1, t1, d(hw), d(hb))  # no SSFA output: same head bits
    assert torch.equal(got, again)
    two = ops.ssfa_fuse(d(x0), d(x1), d(w0), d(w1), s0, t0, s1, t1)
    assert float((two - out).abs().max()) < 1e-6 * max(1.0, float(ssfa.abs().max()))   # the same blend (fma contraction may differ)
    nob = ops.ssfa_fuse_head(d(x0), d(x1), d(w0), d(w1), s0, t0, s1, t1, d(hw), None)
    assert float((nob.cpu().double().view(B, 22, H, W) - (head - hb.double().view(1, -1, 1, 1))).abs().max()) < 2e-5 * max(1.0, float(head.abs().max()))


def test_full_size_layer_bench(dev):
    """KITTI-size 3x3 128->128 @200x176: correctness on a strided sample + a timing printout per tile cfg."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 128, 200, 176, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    ref = F.conv2d(x, w, padding=1)
    xd = x.to(dev)
    pc = ops.pack_conv2d(w.to(dev))
    for cfg in (0, 1, 2, 3):
        out = ops.conv2d(xd, pc, None, None, False, tile_cfg=cfg)
        torch.cuda.synchronize()
        _close(out.cpu(), ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d(xd, pc, None, None, False, out=out, tile_cfg=cfg)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("conv3x3 128->128 @200x176 cfg %d: %.3f ms  %.1f TFLOP/s" % (cfg, ms, 10.38e9 / ms / 1e9))


def _taps_ref(w_co_ci_t):
    """[cin/2][ntaps][2][cout_pad32] from (cout, cin, ntaps), as plain torch indexing"""
    co, ci, nt = w_co_ci_t.shape
    cp = (co + 31) // 32 * 32
    out = torch.zeros((ci // 2, nt, 2, cp), dtype=torch.float32, device=w_co_ci_t.device)
    out[:, :, :, :co] = w_co_ci_t.permute(1, 2, 0).reshape(ci // 2, 2, nt, co).permute(0, 2, 1, 3)
    return out


@pytest.mark.parametrize("co,ci", [(40, 16), (128, 128), (22, 32), (200, 64)])
def test_device_weight_packers_equal_the_torch_restatement(dev, co, ci):
    """sessd_conv2d_pack_taps / sessd_conv3x3_winograd_pack (one launch per packing) vs the torch permute / stack / einsum chains they
    replace: direct layout, the adjoint (data-gradient) layer, the four tap classes of the transposed conv, the three Winograd layouts."""
    g = torch.Generator().manual_seed(co + ci)
    w = torch.randn(co, ci, 3, 3, generator=g).to(dev)
    pc = ops.pack_conv2d(w, 1)
    assert torch.equal(pc.launches[0]["wpk"], _taps_ref(w.reshape(co, ci, 9)))
    if co % 2 == 0:
        wd = w.flip(2, 3).transpose(0, 1).contiguous()          # (ci, co, 3, 3): the adjoint layer's weight
        pa = ops.pack_conv2d(w, 1, adjoint=True)
        assert (pa.cin, pa.cout) == (co, ci) and torch.equal(pa.launches[0]["wpk"], _taps_ref(wd.reshape(ci, co, 9)))
    w1 = torch.randn(co, ci, 1, 1, generator=g).to(dev)
    assert torch.equal(ops.pack_conv2d(w1).launches[0]["wpk"], _taps_ref(w1.reshape(co, ci, 1)))
    # transposed conv: weight (Cin, Cout, 3, 3)
    wt = torch.randn(ci, co, 3, 3, generator=g).to(dev)
    pd = ops.pack_deconv2d_s2(wt)
    sel = {0: [(1, 0)], 1: [(0, 1), (2, 0)]}
    k = 0
    for py in (0, 1):
        for px in (0, 1):
            taps = [(ky, kx) for (ky, _) in sel[py] for (kx, _) in sel[px]]
            ref = _taps_ref(torch.stack([wt[:, :, ky, kx] for ky, kx in taps], -1).permute(1, 0, 2).contiguous())
            assert torch.equal(pd.launches[k]["wpk"], ref), (py, px)
            k += 1
    # Winograd: U = G g G^T (float64, rounded once) in the three layouts
    for adjoint in ((False, True) if co % 2 == 0 else (False,)):
        wv = w.flip(2, 3).transpose(0, 1).contiguous() if adjoint else w
        o, c = wv.shape[0], wv.shape[1]
        U = ops.winograd_u(wv)                                  # (o, c, 16)
        ref0 = _taps_ref(U.contiguous())
        ref0 = ref0.view(c // 2, 4, 4, 2, ref0.shape[3]).permute(0, 1, 3, 4, 2).contiguous()
        got0 = ops.pack_winograd(w, adjoint=adjoint)
        assert got0.shape == ref0.shape and float((got0 - ref0).abs().max()) <= 1.2e-7 * float(U.abs().max())
        for shape, (nw, cc) in enumerate(((8, 128), (4, 64))):
            ng = (o + cc - 1) // cc
            Up = torch.zeros((ng * cc, c, 16), dtype=torch.float32, device=dev)
            Up[:o] = U
            ref = Up.view(ng, cc // 32, 32, c // 2, 2, nw, 16 // nw).permute(0, 3, 5, 4, 2, 1, 6).contiguous()
            got = ops.pack_winograd_sk(w, shape, adjoint=adjoint)
            assert got.shape == ref.shape and float((got - ref).abs().max()) <= 1.2e-7 * float(U.abs().max())
