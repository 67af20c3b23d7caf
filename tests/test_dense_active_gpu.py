"""Active-tile mode of the first SSFA layers (csrc/dense_active.hip + conv3x3s1_winograd_sk_kernel<.., LIST>): on a BEV map that is
zero outside the sparse sites (scn.py:179-183 `.dense()`), the three conv + BatchNorm + ReLU layers of rpn_v1.py:135-148 are
computed only where they are not constant. Checks: the tile masks / lists against a numpy restatement of the rule, and the chain
fill + active conv against the DENSE kernels on the same input (float32 rounding: 1e-5 of the layer's largest value; the first
layer's filled tiles bit-equal)."""
import numpy as np
import pytest
import torch

from sessd_hip import ops

pytestmark = pytest.mark.gpu
H, W = 200, 176


def _sites(seed, batch, n):
    rng = np.random.RandomState(seed)
    rows = []
    for b in range(batch):
        # clustered like a scan: a few blobs + scattered pixels, two z slices
        cy, cx = rng.randint(10, H - 10, 12), rng.randint(10, W - 10, 12)
        y = np.clip(np.concatenate([rng.normal(cy[i], 4, n // 16) for i in range(12)] + [rng.randint(0, H, n // 4)]).astype(int), 0, H - 1)
        x = np.clip(np.concatenate([rng.normal(cx[i], 6, n // 16) for i in range(12)] + [rng.randint(0, W, n // 4)]).astype(int), 0, W - 1)
        z = rng.randint(0, 2, len(y))
        u = np.unique(np.stack([np.full(len(y), b), z, y, x], 1), axis=0)
        rows.append(u)
    return np.concatenate(rows).astype(np.int32)


def _masks_numpy(idx, batch, steps):
    """The rule of csrc/dense_active.hip restated. Step 0 (3x3 stride-1 layer): tile computed iff its 4x4 input patch touches a
    non-constant pixel (+ the border ring once the input constant is not zero, i.e. from the second layer on); the layer's output
    is non-constant exactly in its computed tiles. Step 1 (3x3 stride-2 layer, padding 1, computed everywhere): output pixel
    non-constant iff its 3x3 window is (+ the top row / left column, where the padding enters). Step 2: the same layer, taking a
    slot of its own -- the 2x2 tiles of its output that hold a non-constant pixel (the next layer still sees the pixels). Step 3
    (stride-2 transposed conv on the current map, residual = a map of the last layer slot before the halving): 2x2 tile of its
    INPUT computed iff input rows 2ty .. 2ty+2, cols 2tx .. 2tx+2 hold a non-constant pixel, or it is in the last tile row /
    column (once the constant is not zero), or one of the 2x2 kept tiles it covers was computed. Returns per image the list of
    the slots' tile masks (flattened)."""
    steps = [0] * steps if isinstance(steps, int) else steps
    out = []
    for b in range(batch):
        nc = np.zeros((H, W), bool)
        s = idx[idx[:, 0] == b]
        nc[s[:, 2], s[:, 3]] = True
        per, zero_input = [], True
        last_layer, keep = None, None
        for k in steps:
            h, w = nc.shape
            if k == 4:   # the output of the transposed conv in front: the 4x4 blocks of its tiles, twice the resolution
                nc = per[-1].reshape(h // 2, w // 2).repeat(4, 0).repeat(4, 1)
                zero_input, last_layer, keep = False, None, None
                continue
            p = np.pad(nc, 1)
            if k == 3:
                q = np.pad(nc, ((0, 2), (0, 2)))
                tm = np.zeros((h // 2, w // 2), bool)
                for dy in range(3):
                    for dx in range(3):
                        tm |= q[dy:dy + h:2, dx:dx + w:2][:h // 2, :w // 2]
                if not zero_input:
                    tm[-1, :] = True
                    tm[:, -1] = True
                if keep is not None:
                    tm |= keep.reshape(h // 2, 2, w // 2, 2).any((1, 3))
                per.append(tm.reshape(-1).copy())
                last_layer = None
                continue
            if k != 0:
                keep, last_layer = last_layer, None
            if k == 0:
                tm = np.zeros((h // 2, w // 2), bool)
                for dy in range(4):
                    for dx in range(4):
                        tm |= p[dy:dy + h:2, dx:dx + w:2][:h // 2, :w // 2]
                if not zero_input:
                    tm[0, :] = tm[-1, :] = True
                    tm[:, 0] = tm[:, -1] = True
                per.append(tm.reshape(-1).copy())
                last_layer = tm.copy()
                nc = tm.repeat(2, 0).repeat(2, 1)
                zero_input = False
            else:
                o = np.zeros((h // 2, w // 2), bool)
                for dy in range(3):
                    for dx in range(3):
                        o |= p[dy:dy + h:2, dx:dx + w:2][:h // 2, :w // 2]
                if not zero_input:
                    o[0, :] = True
                    o[:, 0] = True
                if k == 2:
                    per.append(o.reshape(h // 4, 2, w // 4, 2).any((1, 3)).reshape(-1).copy())
                nc = o
                zero_input = False
        out.append(per)
    return out


@pytest.mark.parametrize("batch,steps", [(1, 3), (3, 3), (1, [0, 0, 0, 1, 0, 0]), (2, [0, 0, 0, 1, 0, 0]), (1, [0, 0, 0, 2, 0, 0]),
                                         (3, [0, 0, 0, 2, 0, 0]), (2, [2, 0]), (1, [0, 0, 0, 2, 0, 0, 3]), (3, [0, 0, 0, 2, 0, 0, 3]),
                                         (2, [0, 1, 3]), (1, [2, 3]), (1, [0, 0, 0, 2, 0, 0, 3, 4, 0]), (2, [0, 0, 0, 2, 0, 0, 3, 4, 0]),
                                         (1, [2, 3, 4, 0])])
def test_tile_masks_and_lists(dev, batch, steps):
    idx = _sites(1, batch, 1600 if isinstance(steps, int) else 500)
    ta = ops.TileActivity(batch, H, W, steps, dev)
    n = torch.tensor([len(idx)], dtype=torch.int32, device=dev)
    cap = len(idx) + 100
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    buf[:len(idx)] = torch.from_numpy(idx).to(dev)
    ta.run(buf, n, cap)
    want = _masks_numpy(idx, batch, steps)
    tl, nl = ta.tile_list.cpu().numpy(), ta.n_list.cpu().numpy()
    for l in range(ta.n_slots):
        h, w = ta.dims[l]
        tiles = (h // 2) * (w // 2)
        ref = np.concatenate([np.nonzero(want[b][l])[0] + b * tiles for b in range(batch)])
        tm = ta.mask_bool(l).numpy().reshape(batch, -1)
        for b in range(batch):
            assert np.array_equal(tm[b], want[b][l]), (l, b)
        assert nl[l] == len(ref)
        assert np.array_equal(tl[l, :nl[l]], ref)     # ascending (image, tile): deterministic order
    st = [0] * steps if isinstance(steps, int) else steps
    if ta.n_slots >= 3 and st[:3] == [0, 0, 0]:
        assert 0.05 < nl[0] / (batch * (H // 2) * (W // 2)) < 0.6 and nl[0] < nl[1] < nl[2]
    if ta.n_slots == 5:
        assert ta.dims[3] == (H // 2, W // 2) and nl[3] < nl[4] <= batch * (H // 4) * (W // 4)
    if st == [2, 3, 4, 0]:   # stride-2 slot, the pair's slot, conv_0 / conv_1 behind the pair: back on the first resolution
        assert ta.dims == [(H // 2, W // 2), (H // 2, W // 2), (H, W)] and 4 * nl[1] <= nl[2] <= batch * (H // 2) * (W // 2)
    if ta.n_slots >= 7:   # the transposed convs' slot: at least what the layer before them computed
        assert ta.dims[6] == (H // 2, W // 2) and nl[5] <= nl[6] <= batch * (H // 4) * (W // 4)
    if ta.n_slots == 8:   # conv_0 / conv_1 behind them (steps 4, 0): back on the first resolution, at least the 4 tiles of every pair tile
        assert ta.dims[7] == (H, W) and 4 * nl[6] <= nl[7] <= batch * (H // 2) * (W // 2)
    if ta.n_slots >= 6 and st[:6] == [0, 0, 0, 2, 0, 0]:   # the stride-2 layer's own slot: fewer tiles than the layer after it computes
        assert ta.dims[3] == ta.dims[4] == (H // 2, W // 2) and nl[3] < nl[4] < nl[5] <= batch * (H // 4) * (W // 4)


def _layer(seed, c):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(c, c, 3, 3, generator=g) * (1.0 / (3.0 * c ** 0.5))
    scale = 0.5 + torch.rand(c, generator=g)
    shift = torch.randn(c, generator=g) * 0.3
    return w, scale, shift


def _torch_cbr(x, w, scale, shift, stride=1, residual=None, transposed=False):
    """FIRST-HAND reference of one neck layer (round-4 review item: the list kernels were only held to the repo's own dense
    kernels): torch on the CPU in float64 -- conv (rpn_v1.py:135-160: 3x3 pad 1 / 1x1 pad 0) or ConvTranspose2d(3, stride 2,
    padding 1, output_padding 1) (:176-199), folded BatchNorm, ReLU, then the residual (:224) -- rounded once to float32."""
    import torch.nn.functional as F
    xd, wd = x.detach().cpu().double(), w.detach().cpu().double()
    if transposed:
        y = F.conv_transpose2d(xd, wd, stride=2, padding=1, output_padding=1)
    else:
        y = F.conv2d(xd, wd, stride=stride, padding=wd.shape[-1] // 2)
    y = torch.relu(y * scale.detach().cpu().double().view(1, -1, 1, 1) + shift.detach().cpu().double().view(1, -1, 1, 1))
    if residual is not None:
        y = y + residual.detach().cpu().double()
    return y.float()


def _check_first_hand(got, ref, what, tol=2e-5):
    """device output against the float64 torch reference of THE SAME input: computed tiles (the list kernel's arithmetic) and
    filled tiles (the host's float64 constants) alike, tol * the layer's largest value"""
    r, e = float(ref.abs().max()), float((got.cpu() - ref).abs().max())
    assert e <= tol * r, (what, e, r)


def _constants(layers):
    """value of a layer's output where its input is the previous layer's constant (float64, then float32)"""
    c = torch.zeros(layers[0][0].shape[1], dtype=torch.float64)
    out = []
    for w, scale, shift in layers:
        c = torch.relu(scale.double() * (w.double().sum((2, 3)) @ c) + shift.double())
        out.append(c.float())
    return out


@pytest.mark.parametrize("batch,shape,min_rounds", [(1, 0, 2), (1, 1, 2), (2, 0, 1), (2, 1, 4), (1, 0, 8), (1, 1, -1), (2, 0, -1), (3, 1, -3)])
def test_active_chain_equals_the_dense_layers(dev, batch, shape, min_rounds):
    """min_rounds < 0 (round 5): whole-unit shares -- no unit is cut, so the partial-sum scratch of the workspace is never written"""
    C = 128
    idx = _sites(2 + batch, batch, 1600)
    x = torch.zeros(batch, C, H, W)
    g = torch.Generator().manual_seed(7)
    vals = torch.randn(len(idx), C // 2, generator=g)
    for (b, z, y, xx), v in zip(idx, vals):
        x[b, z * (C // 2):(z + 1) * (C // 2), y, xx] = v
    x = x.to(dev)
    layers = [_layer(10 + l, C) for l in range(3)]
    consts = [c.to(dev) for c in _constants(layers)]
    ta = ops.TileActivity(batch, H, W, 3, dev)
    n = torch.tensor([len(idx)], dtype=torch.int32, device=dev)
    ta.run(torch.from_numpy(idx).to(dev), n, len(idx))
    outs = [torch.full((batch, C, H, W), float("nan"), device=dev) for _ in range(3)]
    ta.fill(outs, consts)
    ws = torch.zeros(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(batch, H, W, C, shape, 0)), dtype=torch.uint8, device=dev)
    ws_list = torch.zeros_like(ws)   # the list launches' own workspace: whole-unit shares must leave ALL of it untouched
    cur_a, cur_d = x, x
    for l, (w, scale, shift) in enumerate(layers):
        pc = ops.pack_conv2d(w.to(dev))
        sc, sh = scale.to(dev), shift.to(dev)
        dense = ops.conv2d(cur_d, pc, sc, sh, True, None, None, 22 + shape, workspace=ws)
        ops.conv2d_winograd_sk_active(cur_a, pc.upk_sk(shape), C, sc, sh, True, outs[l], shape, ws_list, ta.tile_list[l], ta.n_list[l:l + 1],
                                      min_rounds=min_rounds)
        torch.cuda.synchronize()
        if min_rounds < 0:
            assert int(ws_list.view(torch.int32).abs().sum()) == 0, "a whole-unit launch wrote a partial sum"
        got = outs[l]
        assert torch.isfinite(got).all(), "layer %d: a tile neither filled nor computed" % l
        # first hand: torch float64 on the very input the list launch read (its computed tiles AND the filled constants)
        _check_first_hand(got, _torch_cbr(cur_a, w, scale, shift), "Winograd list layer %d vs torch" % l)
        ref = float(dense.abs().max())
        err = float((got - dense).abs().max())
        assert err <= 1e-5 * ref, (l, err, ref)
        if l == 0:
            # filled tiles of the first layer: relu(shift) exactly, as 0 * U gives
            tm = ta.mask_bool(0).to(dev).view(batch, 1, H // 2, W // 2).repeat_interleave(2, 2).repeat_interleave(2, 3).expand(-1, C, -1, -1)
            assert torch.equal(got[~tm], dense[~tm])
        cur_a, cur_d = got, dense
    # the workspace counters are left zero (the next launch relies on it)
    units = batch * ((H // 2) * (W // 2) + 31) // 32 * (C // (128 if shape == 0 else 64))
    assert int(ws[:units * 4].view(torch.int32).abs().sum()) == 0 and int(ws_list[:units * 4].view(torch.int32).abs().sum()) == 0


def test_active_chain_through_the_stride_2_layer(dev):
    """rpn_v1.py:135-160 as the engine runs it: three active layers at 200 x 176, the stride-2 layer over the whole map, two active
    layers of 256 channels at 100 x 88 -- against the dense kernels, 2e-5 of each layer's largest value."""
    batch, C0, C1 = 1, 128, 256
    idx = _sites(21, batch, 400)
    x = torch.zeros(batch, C0, H, W)
    x[idx[:, 0], :, idx[:, 2], idx[:, 3]] = torch.randn(len(idx), C0, generator=torch.Generator().manual_seed(5))
    x = x.to(dev)
    g = torch.Generator().manual_seed(11)
    def mk(ci, co):
        return (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5), 0.5 + torch.rand(co, generator=g), torch.randn(co, generator=g) * 0.3)
    layers = [mk(C0, C0), mk(C0, C0), mk(C0, C0), mk(C0, C1), mk(C1, C1), mk(C1, C1)]
    steps = [0, 0, 0, 1, 0, 0]
    c = torch.zeros(C0, dtype=torch.float64)
    consts = []
    for w_, sc_, sh_ in layers:
        c = torch.relu(sc_.double() * (w_.double().sum((2, 3)) @ c) + sh_.double())
        consts.append(c.float().to(dev))
    ta = ops.TileActivity(batch, H, W, steps, dev)
    ta.run(torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev), len(idx))
    outs = [torch.full((batch, C0, H, W), float("nan"), device=dev) for _ in range(3)] + \
           [torch.full((batch, C1, H // 2, W // 2), float("nan"), device=dev) for _ in range(2)]
    ta.fill(outs, [consts[0], consts[1], consts[2], consts[4], consts[5]])
    ws = torch.zeros(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(batch, H, W, C1, 1, 0)), dtype=torch.uint8, device=dev)
    cur_a = cur_d = x
    slot = 0
    for li, (w_, sc_, sh_) in enumerate(layers):
        sc, sh = sc_.to(dev), sh_.to(dev)
        if li == 3:   # the stride-2 layer: dense on both sides
            pc = ops.pack_conv2d(w_.to(dev), 2)
            cur_a, cur_d = ops.conv2d(cur_a, pc, sc, sh, True), ops.conv2d(cur_d, pc, sc, sh, True)
            continue
        pc = ops.pack_conv2d(w_.to(dev))
        dense = ops.conv2d(cur_d, pc, sc, sh, True, None, None, 23)
        ops.conv2d_winograd_sk_active(cur_a, pc.upk_sk(1), pc.cout, sc, sh, True, outs[slot], 1, ws, ta.tile_list[slot], ta.n_list[slot:slot + 1])
        got = outs[slot]
        assert torch.isfinite(got).all(), "layer %d: a tile neither filled nor computed" % li
        ref, err = float(dense.abs().max()), float((got - dense).abs().max())
        assert err <= 2e-5 * ref, (li, err, ref)
        cur_a, cur_d = got, dense
        slot += 1
    frac = [float(ta.n_list[s]) / (batch * (ta.dims[s][0] // 2) * (ta.dims[s][1] // 2)) for s in range(5)]
    print("computed tile fractions", [round(f, 3) for f in frac])
    assert frac[0] < frac[1] < frac[2] and frac[3] < frac[4] < 1.0


@pytest.mark.parametrize("batch,min_rounds", [(1, 1), (1, 4), (2, 2), (3, 8)])
def test_stride_2_and_1x1_layers_over_tile_lists(dev, batch, min_rounds):
    """rpn_v1.py:150-152,163-172 in active-tile mode (conv2d_sk_kernel<LIST>, sessd_conv2d_sk_active): the stride-2 conv that opens
    block 1 over ITS tile list (step 2 of the activity program), the 1x1 trans layers over the list of the layer that produced
    their input -- against the dense stream-K launches on the same inputs. A computed pixel is the dense kernel's arithmetic in
    another summation split (stream-K shares differ): 1e-5 of the layer's largest value; filled pixels: the constants' chain."""
    C0, C1 = 128, 256
    idx = _sites(31 + batch, batch, 500)
    x = torch.zeros(batch, C0, H, W)
    x[idx[:, 0], :, idx[:, 2], idx[:, 3]] = torch.randn(len(idx), C0, generator=torch.Generator().manual_seed(5))
    x = x.to(dev)
    g = torch.Generator().manual_seed(13)
    def mk(ci, co, k):
        return (torch.randn(co, ci, k, k, generator=g) / (k * ci ** 0.5), 0.5 + torch.rand(co, generator=g), torch.randn(co, generator=g) * 0.3)
    l0, l1, l2, tr0, tr1 = mk(C0, C0, 3), mk(C0, C1, 3), mk(C1, C1, 3), mk(C0, C0, 1), mk(C1, C1, 1)
    def const(layer, c):
        w_, sc_, sh_ = layer
        return torch.relu(sc_.double() * (w_.double().sum((2, 3)) @ c) + sh_.double())
    c0 = const(l0, torch.zeros(C0, dtype=torch.float64)); c1 = const(l1, c0); c2 = const(l2, c1)
    ct0, ct1 = const(tr0, c0), const(tr1, c2)
    f32 = lambda v: v.float().to(dev)
    ta = ops.TileActivity(batch, H, W, [0, 2, 0], dev)
    ta.run(torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev), len(idx))
    nan = lambda c, h, w: torch.full((batch, c, h, w), float("nan"), device=dev)
    o0, o1, o2, ot0, ot1 = nan(C0, H, W), nan(C1, H // 2, W // 2), nan(C1, H // 2, W // 2), nan(C0, H, W), nan(C1, H // 2, W // 2)
    ta.fill([o0, o1, o2, ot0, ot1], [f32(c0), f32(c1), f32(c2), f32(ct0), f32(ct1)], layers=[0, 1, 2, 0, 2])
    ws = torch.zeros(max(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(batch, H, W, C1, 1, 0)),
                         int(ops.lib.sessd_conv2d_sk_workspace_bytes(batch, H, W, C1, 1, 0))), dtype=torch.uint8, device=dev)
    dv = lambda layer: (layer[1].to(dev), layer[2].to(dev))
    def check(got, dense, what, tol=1e-5):
        assert torch.isfinite(got).all(), "%s: a pixel neither filled nor computed" % what
        ref, err = float(dense.abs().max()), float((got - dense).abs().max())
        assert err <= tol * ref, (what, err, ref)
    # layer 0 (3x3 stride 1, Winograd list kernel) gives the input of the stride-2 layer and of trans_0
    p0 = ops.pack_conv2d(l0[0].to(dev))
    d0 = ops.conv2d(x, p0, *dv(l0), True, None, None, 23)
    ops.conv2d_winograd_sk_active(x, p0.upk_sk(1), C0, *dv(l0), True, o0, 1, ws, ta.tile_list[0], ta.n_list[0:1])
    check(o0, d0, "layer 0")
    # the stride-2 layer over its own list
    p1 = ops.pack_conv2d(l1[0].to(dev), 2)
    d1 = ops.conv2d(d0, p1, *dv(l1), True, None, None, 30, workspace=ws)
    ops.conv2d_sk_active(o0, p1, *dv(l1), True, o1, ws, ta.tile_list[1], ta.n_list[1:2], min_rounds=min_rounds)
    check(o1, d1, "stride-2 layer", 2e-5)
    _check_first_hand(o1, _torch_cbr(o0, l1[0], l1[1], l1[2], stride=2), "stride-2 list layer (conv2d_sk LIST) vs torch")
    # the layer after it (Winograd list kernel) sees the pixel-level rows of the transition
    p2 = ops.pack_conv2d(l2[0].to(dev))
    d2 = ops.conv2d(d1, p2, *dv(l2), True, None, None, 23)
    ops.conv2d_winograd_sk_active(o1, p2.upk_sk(1), C1, *dv(l2), True, o2, 1, ws, ta.tile_list[2], ta.n_list[2:3])
    check(o2, d2, "layer 2", 2e-5)
    # the 1x1 layers: computed where their input was
    pt0, pt1 = ops.pack_conv2d(tr0[0].to(dev)), ops.pack_conv2d(tr1[0].to(dev))
    dt0 = ops.conv2d(d0, pt0, *dv(tr0), True, None, None, 30, workspace=ws)
    dt1 = ops.conv2d(d2, pt1, *dv(tr1), True, None, None, 30, workspace=ws)
    ops.conv2d_sk_active(o0, pt0, *dv(tr0), True, ot0, ws, ta.tile_list[0], ta.n_list[0:1], min_rounds=min_rounds)
    ops.conv2d_sk_active(o2, pt1, *dv(tr1), True, ot1, ws, ta.tile_list[2], ta.n_list[2:3], min_rounds=min_rounds)
    check(ot0, dt0, "trans_0", 2e-5)
    check(ot1, dt1, "trans_1", 2e-5)
    _check_first_hand(ot0, _torch_cbr(o0, *tr0), "trans_0 (conv2d_sk LIST, 1x1) vs torch")
    _check_first_hand(ot1, _torch_cbr(o2, *tr1), "trans_1 (conv2d_sk LIST, 1x1) vs torch")
    # a residual rides along in the listed pixels only (the engine does not use one here; the entry point takes it)
    res = torch.randn(batch, C0, H, W, device=dev)
    keep = ot0.clone()
    ops.conv2d_sk_active(o0, pt0, *dv(tr0), True, ot0, ws, ta.tile_list[0], ta.n_list[0:1], residual=res, min_rounds=min_rounds)
    tm = ta.mask_bool(0).to(dev).view(batch, 1, H // 2, W // 2).repeat_interleave(2, 2).repeat_interleave(2, 3).expand(-1, C0, -1, -1)
    assert torch.equal(ot0[~tm], keep[~tm]) and torch.allclose(ot0[tm], (keep + res)[tm], rtol=0, atol=1e-6 * float(keep.abs().max()) + 1e-6)
    # same list, same launch configuration -> same bits; the counters are left zero
    again = nan(C1, H // 2, W // 2)
    ta.fill([again], [f32(c1)], layers=[1])
    ops.conv2d_sk_active(o0, p1, *dv(l1), True, again, ws, ta.tile_list[1], ta.n_list[1:2], min_rounds=min_rounds)
    assert torch.equal(again, o1)
    units = batch * ((H * W + 127) // 128) * 2
    assert int(ws[:units * 4].view(torch.int32).abs().sum()) == 0
    frac = [float(ta.n_list[s]) / (batch * (ta.dims[s][0] // 2) * (ta.dims[s][1] // 2)) for s in range(3)]
    print("computed tile fractions", [round(f, 3) for f in frac])
    assert frac[0] < 0.6 and frac[1] < frac[2] < 1.0


@pytest.mark.parametrize("batch,cfg", [(1, 11), (2, 4), (3, 12), (1, 3)])
def test_direct_kernels_over_tile_lists(dev, batch, cfg):
    """rpn_v1.py:163-199, 224 on the direct kernel over lists (conv_body<.., LIST>: sessd_conv2d_mfma_active,
    sessd_deconv2d_s2_mfma_pair_active): trans_0 over the list of the layer that produced its input, the two transposed convs
    over the step-3 slot (2x2 tiles of their input = 4x4 output blocks; residual from the full-resolution layer). A computed pixel
    is the plain launch's code: BIT-EQUAL to it on the same input; a filled pixel: the constants' chain (one value per output
    parity class for the transposed convs), 1e-5 of the layer's largest value."""
    C0, C1 = 128, 256
    idx = _sites(41 + batch, batch, 500)
    x = torch.zeros(batch, C0, H, W)
    x[idx[:, 0], :, idx[:, 2], idx[:, 3]] = torch.randn(len(idx), C0, generator=torch.Generator().manual_seed(5))
    x = x.to(dev)
    g = torch.Generator().manual_seed(17)
    def mk(ci, co, k):
        return (torch.randn(co, ci, k, k, generator=g) / (k * ci ** 0.5), 0.5 + torch.rand(co, generator=g), torch.randn(co, generator=g) * 0.3)
    l0, l1, tr0, tr1 = mk(C0, C0, 3), mk(C0, C1, 3), mk(C0, C0, 1), mk(C1, C1, 1)
    dwa = (torch.randn(C1, C0, 3, 3, generator=g) / (3 * C1 ** 0.5), 0.5 + torch.rand(C0, generator=g), torch.randn(C0, generator=g) * 0.3)
    dwb = (torch.randn(C1, C0, 3, 3, generator=g) / (3 * C1 ** 0.5), 0.5 + torch.rand(C0, generator=g), torch.randn(C0, generator=g) * 0.3)
    def const(layer, c):
        w_, sc_, sh_ = layer
        return torch.relu(sc_.double() * (w_.double().sum((2, 3)) @ c) + sh_.double())
    def dconst(layer, c):
        w_, sc_, sh_ = layer
        K = {0: [1], 1: [0, 2]}
        return torch.stack([torch.relu(sc_.double() * (c @ sum(w_.double()[:, :, ky, kx] for ky in K[py] for kx in K[px])) + sh_.double())
                            for py in (0, 1) for px in (0, 1)])
    c0 = const(l0, torch.zeros(C0, dtype=torch.float64)); c1 = const(l1, c0)
    ct0, ct1 = const(tr0, c0), const(tr1, c1)
    cda, cdb = dconst(dwa, ct1) + ct0[None], dconst(dwb, ct1)
    f32 = lambda v: v.float().to(dev).contiguous()
    dv = lambda layer: (layer[1].to(dev), layer[2].to(dev))
    ta = ops.TileActivity(batch, H, W, [0, 2, 3], dev)
    ta.run(torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev), len(idx))
    # the inputs through the dense kernels (the list kernels of the layers before are tested above)
    p0, p1 = ops.pack_conv2d(l0[0].to(dev)), ops.pack_conv2d(l1[0].to(dev), 2)
    x0 = ops.conv2d(x, p0, *dv(l0), True, None, None, 20)
    x1 = ops.conv2d(x0, p1, *dv(l1), True)
    pt0, pt1 = ops.pack_conv2d(tr0[0].to(dev)), ops.pack_conv2d(tr1[0].to(dev))
    d_t0 = ops.conv2d(x0, pt0, *dv(tr0), True, None, None, cfg)
    d_t1 = ops.conv2d(x1, pt1, *dv(tr1), True, None, None, cfg)
    pa, pb = ops.pack_deconv2d_s2(dwa[0].to(dev)), ops.pack_deconv2d_s2(dwb[0].to(dev))
    d_a, d_b = torch.empty(batch, C0, H, W, device=dev), torch.empty(batch, C0, H, W, device=dev)
    ops.deconv2d_s2_pair(d_t1, pa, pb, *dv(dwa), *dv(dwb), True, d_a, d_b, residual_a=d_t0, tile_cfg=cfg)
    nan = lambda c, h, w: torch.full((batch, c, h, w), float("nan"), device=dev)
    o_t0, o_t1, o_a, o_b = nan(C0, H, W), nan(C1, H // 2, W // 2), nan(C0, H, W), nan(C0, H, W)
    # slot 0 = layer 0's tiles (trans_0's list), slot 1 = the stride-2 layer's own tiles (trans_1's list here), slot 2 = the
    # transposed convs' input tiles
    ta.fill([o_t0, o_t1, o_a, o_b], [f32(ct0), f32(ct1), f32(cda), f32(cdb)], layers=[0, 1, 2, 2], tiles=[2, 2, 4, 4])
    ops.conv2d_mfma_active(x0, pt0, *dv(tr0), True, o_t0, ta.tile_list[0], ta.n_list[0:1], tile_cfg=cfg)
    ops.conv2d_mfma_active(x1, pt1, *dv(tr1), True, o_t1, ta.tile_list[1], ta.n_list[1:2], tile_cfg=cfg)
    ops.deconv2d_s2_pair_active(d_t1, pa, pb, *dv(dwa), *dv(dwb), True, o_a, o_b, ta.tile_list[2], ta.n_list[2:3], residual_a=d_t0, tile_cfg=cfg)
    torch.cuda.synchronize()
    def tmask(slot, up):
        h, w = ta.dims[slot]
        return ta.mask_bool(slot).to(dev).view(batch, 1, h // 2, w // 2).repeat_interleave(2 * up, 2).repeat_interleave(2 * up, 3)
    for got, dense, slot, up, what in ((o_t0, d_t0, 0, 1, "trans_0"), (o_t1, d_t1, 1, 1, "trans_1"), (o_a, d_a, 2, 2, "deconv_0 + residual"),
                                       (o_b, d_b, 2, 2, "deconv_1")):
        assert torch.isfinite(got).all(), "%s: a pixel neither filled nor computed" % what
        tm = tmask(slot, up).expand_as(got)
        assert torch.equal(got[tm], dense[tm]), "%s: a computed pixel differs from the plain launch" % what
        ref, err = float(dense.abs().max()), float((got - dense).abs().max())
        assert err <= 1e-5 * ref, (what, err, ref)
    # first hand: torch float64 on the inputs the list launches read
    _check_first_hand(o_t0, _torch_cbr(x0, *tr0), "trans_0 (direct 1x1 over a list) vs torch")
    _check_first_hand(o_t1, _torch_cbr(x1, *tr1), "trans_1 (direct 1x1 over a list) vs torch")
    _check_first_hand(o_a, _torch_cbr(d_t1, *dwa, residual=d_t0, transposed=True), "deconv_0 + residual (pair over a list) vs torch")
    _check_first_hand(o_b, _torch_cbr(d_t1, *dwb, transposed=True), "deconv_1 (pair over a list) vs torch")
    frac = [float(ta.n_list[s]) / (batch * (ta.dims[s][0] // 2) * (ta.dims[s][1] // 2)) for s in range(3)]
    print("computed tile fractions", [round(f, 3) for f in frac])
    assert frac[0] < 0.6 and frac[2] < 1.0


@pytest.mark.parametrize("batch,shape,min_rounds", [(1, 0, 8), (1, 1, 4), (2, 0, -1), (1, 1, -1)])
def test_convs_behind_the_transposed_pair_over_their_tile_list(dev, batch, shape, min_rounds):
    """rpn_v1.py:200-210 (conv_0 / conv_1) over the tile list of step program {.., 3, 4, 0} (round 6): the inputs are the transposed
    convs' outputs, constant PER OUTPUT PARITY CLASS outside the 4x4 blocks of the pair's tiles; a 2x2-output tile of the conv is
    computed iff its 4x4 patch touches such a block (or lies on the border ring), the rest of its output is filled with the layer's
    own parity-class constants (sessd_fill_tiles_job_t.tile = 6). Computed pixels against the full-map launch of the same kernel
    family, everything against torch float64 on the same input."""
    C0, C1 = 128, 256
    idx = _sites(71 + batch, batch, 400)
    x = torch.zeros(batch, C0, H, W)
    x[idx[:, 0], :, idx[:, 2], idx[:, 3]] = torch.randn(len(idx), C0, generator=torch.Generator().manual_seed(6))
    x = x.to(dev)
    g = torch.Generator().manual_seed(23)
    def mk(ci, co, k):
        return (torch.randn(co, ci, k, k, generator=g) / (k * ci ** 0.5), 0.5 + torch.rand(co, generator=g), torch.randn(co, generator=g) * 0.3)
    l0, l1, tr0, tr1 = mk(C0, C0, 3), mk(C0, C1, 3), mk(C0, C0, 1), mk(C1, C1, 1)
    dwa = (torch.randn(C1, C0, 3, 3, generator=g) / (3 * C1 ** 0.5), 0.5 + torch.rand(C0, generator=g), torch.randn(C0, generator=g) * 0.3)
    dwb = (torch.randn(C1, C0, 3, 3, generator=g) / (3 * C1 ** 0.5), 0.5 + torch.rand(C0, generator=g), torch.randn(C0, generator=g) * 0.3)
    cv0, cv1 = mk(C0, C0, 3), mk(C0, C0, 3)
    def const(layer, c):
        w_, sc_, sh_ = layer
        return torch.relu(sc_.double() * (w_.double().sum((2, 3)) @ c) + sh_.double())
    def dconst(layer, c):
        w_, sc_, sh_ = layer
        K = {0: [1], 1: [0, 2]}
        return torch.stack([torch.relu(sc_.double() * (c @ sum(w_.double()[:, :, ky, kx] for ky in K[py] for kx in K[px])) + sh_.double())
                            for py in (0, 1) for px in (0, 1)])
    def pconst(layer, cpar):
        w_, sc_, sh_ = layer
        return torch.stack([torch.relu(sc_.double() * sum(w_.double()[:, :, ky, kx] @ cpar[((py + ky - 1) & 1) * 2 + ((px + kx - 1) & 1)]
                                                             for ky in range(3) for kx in range(3)) + sh_.double()) for py in (0, 1) for px in (0, 1)])
    c0 = const(l0, torch.zeros(C0, dtype=torch.float64)); c1 = const(l1, c0)
    ct0, ct1 = const(tr0, c0), const(tr1, c1)
    cda, cdb = dconst(dwa, ct1) + ct0[None], dconst(dwb, ct1)
    cc0, cc1 = pconst(cv0, cda), pconst(cv1, cdb)
    f32 = lambda v: v.float().to(dev).contiguous()
    dv = lambda layer: (layer[1].to(dev), layer[2].to(dev))
    ta = ops.TileActivity(batch, H, W, [0, 2, 3, 4, 0], dev)
    assert ta.n_slots == 4 and ta.dims[3] == (H, W)
    ta.run(torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev), len(idx))
    # the inputs of the stage through the dense kernels (their list forms are tested above)
    p0, p1 = ops.pack_conv2d(l0[0].to(dev)), ops.pack_conv2d(l1[0].to(dev), 2)
    x0 = ops.conv2d(x, p0, *dv(l0), True, None, None, 20)
    x1 = ops.conv2d(x0, p1, *dv(l1), True)
    pt0, pt1 = ops.pack_conv2d(tr0[0].to(dev)), ops.pack_conv2d(tr1[0].to(dev))
    d_t0 = ops.conv2d(x0, pt0, *dv(tr0), True, None, None, 4)
    d_t1 = ops.conv2d(x1, pt1, *dv(tr1), True, None, None, 4)
    pa, pb = ops.pack_deconv2d_s2(dwa[0].to(dev)), ops.pack_deconv2d_s2(dwb[0].to(dev))
    mid0, mid1 = torch.empty(batch, C0, H, W, device=dev), torch.empty(batch, C0, H, W, device=dev)
    ops.deconv2d_s2_pair(d_t1, pa, pb, *dv(dwa), *dv(dwb), True, mid0, mid1, residual_a=d_t0, tile_cfg=4)
    pc0, pc1 = ops.pack_conv2d(cv0[0].to(dev)), ops.pack_conv2d(cv1[0].to(dev))
    ws = torch.zeros(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(batch, H, W, C0, shape, 0)), dtype=torch.uint8, device=dev)
    full0 = ops.conv2d(mid0, pc0, *dv(cv0), True, None, None, 22 + shape, workspace=ws)
    full1 = ops.conv2d(mid1, pc1, *dv(cv1), True, None, None, 22 + shape, workspace=ws)
    o0 = torch.full((batch, C0, H, W), float("nan"), device=dev)
    o1 = torch.full((batch, C0, H, W), float("nan"), device=dev)
    ta.fill([o0, o1], [f32(cc0), f32(cc1)], layers=[3, 3], tiles=[6, 6])
    for pc, layer, xin, out in ((pc0, cv0, mid0, o0), (pc1, cv1, mid1, o1)):
        ops.conv2d_winograd_sk_active(xin, pc.upk_sk(shape), C0, *dv(layer), True, out, shape, ws, ta.tile_list[3], ta.n_list[3:4],
                                      min_rounds=min_rounds)
    torch.cuda.synchronize()
    tm = ta.mask_bool(3).to(dev).view(batch, 1, H // 2, W // 2).repeat_interleave(2, 2).repeat_interleave(2, 3)
    for got, full, layer, xin, what in ((o0, full0, cv0, mid0, "conv_0"), (o1, full1, cv1, mid1, "conv_1")):
        assert torch.isfinite(got).all(), "%s: a pixel neither filled nor computed" % what
        ref = float(full.abs().max())
        m = tm.expand_as(got)
        assert float((got[m] - full[m]).abs().max()) <= 2e-6 * ref, "%s: a computed pixel differs from the full-map launch" % what
        assert float((got - full).abs().max()) <= 1e-5 * ref, what       # filled pixels: the parity-class constants
        _check_first_hand(got, _torch_cbr(xin, *layer), "%s over the transposed pair's output (list + parity fill) vs torch" % what)
    frac = float(ta.n_list[3]) / (batch * (H // 2) * (W // 2))
    pair = float(ta.n_list[2]) / (batch * (H // 4) * (W // 4))
    print("computed tile fractions: pair %.3f, convs behind it %.3f" % (pair, frac))
    assert pair < frac < 1.0


@pytest.mark.parametrize("batch", [1, 2])
def test_fill_only_where_the_reader_can_reach(dev, batch):
    """sessd_fill_tiles_job_t.near_mask: a map whose only reader is a 3x3 layer over its own list gets the constant only in the
    tiles that reader can reach. The maps start as NaN: the last layer of a three-layer chain comes out finite and equal to the
    dense chain (nothing unreachable was read), while the intermediate maps keep NaN somewhere (something WAS left out)."""
    C = 128
    idx = _sites(60 + batch, batch, 1200)
    x = torch.zeros(batch, C, H, W)
    x[idx[:, 0], :, idx[:, 2], idx[:, 3]] = torch.randn(len(idx), C, generator=torch.Generator().manual_seed(2))
    x = x.to(dev)
    layers = [_layer(20 + l, C) for l in range(3)]
    consts = [c.to(dev) for c in _constants(layers)]
    ta = ops.TileActivity(batch, H, W, 3, dev)
    ta.run(torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev), len(idx))
    outs = [torch.full((batch, C, H, W), float("nan"), device=dev) for _ in range(3)]
    ta.fill(outs, consts, near=[1, 2, None])
    ws = torch.zeros(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(batch, H, W, C, 1, 0)), dtype=torch.uint8, device=dev)
    cur_a = cur_d = x
    for l, (w, scale, shift) in enumerate(layers):
        pc = ops.pack_conv2d(w.to(dev))
        sc, sh = scale.to(dev), shift.to(dev)
        cur_d = ops.conv2d(cur_d, pc, sc, sh, True, None, None, 23)
        ops.conv2d_winograd_sk_active(cur_a, pc.upk_sk(1), C, sc, sh, True, outs[l], 1, ws, ta.tile_list[l], ta.n_list[l:l + 1])
        cur_a = outs[l]
    torch.cuda.synchronize()
    assert torch.isfinite(outs[2]).all()
    ref, err = float(cur_d.abs().max()), float((outs[2] - cur_d).abs().max())
    assert err <= 2e-5 * ref, (err, ref)
    left = [float(torch.isnan(o).float().mean()) for o in outs[:2]]
    print("share of the maps left alone", [round(v, 3) for v in left])
    assert left[0] > 0.1 and left[1] > 0.02   # (these clustered-plus-scattered sites are far denser in effect than a scan: there 0.68 / 0.58)
    # what was filled or computed is exactly the reach of the reader: 3x3 tiles around its list
    for l in (0, 1):
        reach = torch.nn.functional.max_pool2d(ta.mask_bool(l + 1).float()[:, None], 3, 1, 1)[:, 0] > 0
        mine = ta.mask_bool(l)
        want = (reach | mine).to(dev).view(batch, 1, H // 2, W // 2).repeat_interleave(2, 2).repeat_interleave(2, 3).expand(-1, C, -1, -1)
        assert torch.equal(~torch.isnan(outs[l]), want), l


def test_an_empty_list_computes_nothing(dev):
    """no site -> no tile: the launch returns at once and leaves the output alone"""
    C = 128
    ta = ops.TileActivity(1, H, W, [2], dev)
    idx = torch.zeros((4, 4), dtype=torch.int32, device=dev)
    ta.run(idx, torch.zeros(1, dtype=torch.int32, device=dev), 4)
    assert int(ta.n_list[0]) == 0
    w, scale, shift = _layer(3, C)
    pc = ops.pack_conv2d(w.to(dev), 2)
    ws = torch.zeros(int(ops.lib.sessd_conv2d_sk_workspace_bytes(1, H // 2, W // 2, C, 1, 0)), dtype=torch.uint8, device=dev)
    out = torch.full((1, C, H // 2, W // 2), 3.0, device=dev)
    ops.conv2d_sk_active(torch.randn(1, C, H, W, device=dev), pc, scale.to(dev), shift.to(dev), True, out, ws, ta.tile_list[0], ta.n_list[0:1])
    torch.cuda.synchronize()
    assert bool((out == 3.0).all())


def test_active_layer_is_repeatable(dev):
    """same list, same launch configuration -> same bits (shares are a function of the device count only)"""
    C = 128
    idx = _sites(9, 1, 1600)
    x = torch.zeros(1, C, H, W)
    x[0, :, idx[:, 2], idx[:, 3]] = torch.randn(C, len(idx), generator=torch.Generator().manual_seed(1))
    x = x.to(dev)
    w, scale, shift = _layer(3, C)
    pc = ops.pack_conv2d(w.to(dev))
    ta = ops.TileActivity(1, H, W, 1, dev)
    ta.run(torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev), len(idx))
    ws = torch.zeros(int(ops.lib.sessd_conv3x3_winograd_sk_workspace_bytes(1, H, W, C, 0, 0)), dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(3):
        o = torch.zeros(1, C, H, W, device=dev)
        ops.conv2d_winograd_sk_active(x, pc.upk_sk(0), C, scale.to(dev), shift.to(dev), True, o, 0, ws, ta.tile_list[0], ta.n_list[0:1])
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("batch", [1, 2])
def test_fill_reach_of_readers_on_a_coarser_grid(dev, batch):
    """sessd_fill_tiles_job_t.near_kind (round 5): a map whose list-driven reader works on a tile grid twice as coarse is filled only
    where that reader can touch it. kind 1 = inside the reader's listed tiles (the transposed pair reads its residual there:
    reader tile (Ty, Tx) = tiles (2Ty .. 2Ty+1, 2Tx .. 2Tx+1) of the map); kind 2 = a 3x3 stride-2 layer over 2x2 tiles of its
    OUTPUT (tile (Ty, Tx) reads input pixel rows 4Ty-1 .. 4Ty+3 = tiles 2Ty-1 .. 2Ty+1 of the map). The maps start as NaN; what is
    finite afterwards must be exactly (reach AND NOT computed) -- the engine's end-to-end check with poisoned maps is
    tests/test_pipeline_gpu.py::test_active_tiles_do_not_change_the_frame."""
    C = 32
    idx = _sites(70 + batch, batch, 700)
    ta = ops.TileActivity(batch, H, W, [0, 2, 3], dev)   # slot 0: a layer at 200 x 176; slots 1 / 2: readers on the 50 x 44 grid
    ta.run(torch.from_numpy(idx).to(dev), torch.tensor([len(idx)], dtype=torch.int32, device=dev), len(idx))
    own = ta.mask_bool(0).numpy()                         # (batch, 100, 88)
    for kind, slot in ((2, 1), (1, 2)):
        rd = ta.mask_bool(slot).numpy()                   # (batch, 50, 44)
        th, tw = own.shape[1:]
        reach = np.zeros_like(own)
        for ty in range(th):
            ys = [ty >> 1] + ([(ty >> 1) + 1] if (kind == 2 and ty & 1 and (ty >> 1) + 1 < rd.shape[1]) else [])
            for tx in range(tw):
                xs = [tx >> 1] + ([(tx >> 1) + 1] if (kind == 2 and tx & 1 and (tx >> 1) + 1 < rd.shape[2]) else [])
                reach[:, ty, tx] = np.any([rd[:, y, x] for y in ys for x in xs], axis=0)
        out = torch.full((batch, C, H, W), float("nan"), device=dev)
        val = torch.arange(C, dtype=torch.float32, device=dev) + 1.0
        ta.fill([out], [val], layers=[0], near=[slot], near_kind=[kind])
        torch.cuda.synchronize()
        filled = ~torch.isnan(out)
        want = torch.from_numpy(reach & ~own).to(dev).view(batch, 1, th, tw).repeat_interleave(2, 2).repeat_interleave(2, 3).expand(-1, C, -1, -1)
        assert torch.equal(filled, want), (kind, int((filled != want).sum()))
        assert torch.equal(out[filled], val.view(1, C, 1, 1).expand_as(out)[filled])
        share = float(want.float().mean())
        print("kind", kind, "filled share", round(share, 3), "own", round(float(own.mean()), 3), "reach", round(float(reach.mean()), 3))
        assert 0.02 < share < 0.9
