"""The active-tile RULE itself, held to real convolutions on the CPU (float64): wherever the rule of csrc/dense_active.hip says a
tile is not computed, the dense layer chain of rpn_v1.py:135-199, 224 -- three 3x3 conv + BN + ReLU layers, the stride-2 layer, two
more 3x3 layers, the 1x1 trans layers, the two stride-2 transposed convs with trans_0's map as the residual of the first --
really produces the per-channel constant the engine fills in (one constant per output parity class for the transposed convs).
The GPU tests compare the kernels' masks with this restatement (tests/test_dense_active_gpu.py::_masks_numpy) and the list
launches with the dense kernels; this file pins the restatement and the host-side constants to torch's own convolutions."""
import numpy as np
import torch
import torch.nn.functional as F


def masks(nc0, steps):
    """nc0 (h, w) bool: pixels that hold a site. Returns the slots' tile masks (2-D bool) in step order; the rule of
    csrc/dense_active.hip (steps 0 / 1 / 2 / 3 / 4 as sessd_bev_tile_activity takes them)."""
    nc = nc0.copy()
    per, zero_input, last_layer, keep = [], True, None, None
    for k in steps:
        h, w = nc.shape
        if k == 4:
            # the map becomes the output of the transposed conv in front (the last slot): non-constant = the 4x4 blocks of its tiles
            nc = per[-1].repeat(4, 0).repeat(4, 1)
            zero_input, last_layer, keep = False, None, None
            continue
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
            per.append(tm)
            last_layer = None
            continue
        p = np.pad(nc, 1)
        if k == 0:
            tm = np.zeros((h // 2, w // 2), bool)
            for dy in range(4):
                for dx in range(4):
                    tm |= p[dy:dy + h:2, dx:dx + w:2][:h // 2, :w // 2]
            if not zero_input:
                tm[0, :] = tm[-1, :] = True
                tm[:, 0] = tm[:, -1] = True
            per.append(tm)
            last_layer = tm.copy()
            nc = tm.repeat(2, 0).repeat(2, 1)
        else:
            keep, last_layer = last_layer, None
            o = np.zeros((h // 2, w // 2), bool)
            for dy in range(3):
                for dx in range(3):
                    o |= p[dy:dy + h:2, dx:dx + w:2][:h // 2, :w // 2]
            if not zero_input:
                o[0, :] = True
                o[:, 0] = True
            if k == 2:
                per.append(o.reshape(h // 4, 2, w // 4, 2).any((1, 3)))
            nc = o
        zero_input = False
    return per


def _layer(g, ci, co, k):
    return (torch.randn(co, ci, k, k, generator=g, dtype=torch.float64) / (k * ci ** 0.5),
            0.5 + torch.rand(co, generator=g, dtype=torch.float64), torch.randn(co, generator=g, dtype=torch.float64) * 0.3)


def _cbr(x, layer, stride=1):
    w, s, t = layer
    return torch.relu(F.conv2d(x, w, stride=stride, padding=w.shape[2] // 2) * s[None, :, None, None] + t[None, :, None, None])


def _dbr(x, layer):
    w, s, t = layer   # ConvTranspose2d weight (cin, cout, 3, 3)
    return torch.relu(F.conv_transpose2d(x, w, stride=2, padding=1, output_padding=1) * s[None, :, None, None] + t[None, :, None, None])


def _const(layer, c):
    """what a conv + BN + ReLU layer gives where its whole window holds the constant c (sessd_hip.engine.DensePlan.act_const)"""
    w, s, t = layer
    return torch.relu(s * (w.sum((2, 3)) @ c) + t)


def _dconst(layer, c):
    w, s, t = layer
    K = {0: [1], 1: [0, 2]}
    return torch.stack([torch.relu(s * (c @ sum(w[:, :, ky, kx] for ky in K[py] for kx in K[px])) + t) for py in (0, 1) for px in (0, 1)])


def _up(tm, f):
    return torch.from_numpy(tm.repeat(f, 0).repeat(f, 1))


def test_what_the_rule_leaves_out_is_the_constant():
    H, W, C0, C1 = 96, 112, 6, 8
    rng = np.random.RandomState(3)
    g = torch.Generator().manual_seed(5)
    for trial in range(3):
        nc0 = np.zeros((H, W), bool)
        ys, xs = rng.randint(0, H, 14), rng.randint(0, W, 14)
        nc0[ys, xs] = True
        if trial == 1:
            nc0[0, 0] = nc0[H - 1, W - 1] = nc0[H - 1, 0] = True   # sites in the corners
        if trial == 2:
            nc0[:] = False
            nc0[H // 2, W // 2] = True                            # a single site
        x = torch.zeros(1, C0, H, W, dtype=torch.float64)
        x[0, :, torch.from_numpy(nc0)] = torch.randn(C0, int(nc0.sum()), generator=g, dtype=torch.float64)
        b0 = [_layer(g, C0, C0, 3) for _ in range(3)]
        b1 = [_layer(g, C0, C1, 3), _layer(g, C1, C1, 3), _layer(g, C1, C1, 3)]
        tr0, tr1 = _layer(g, C0, C0, 1), _layer(g, C1, C1, 1)
        da = (torch.randn(C1, C0, 3, 3, generator=g, dtype=torch.float64) / 5, 0.5 + torch.rand(C0, generator=g, dtype=torch.float64),
              torch.randn(C0, generator=g, dtype=torch.float64) * 0.3)
        db = (torch.randn(C1, C0, 3, 3, generator=g, dtype=torch.float64) / 5, 0.5 + torch.rand(C0, generator=g, dtype=torch.float64),
              torch.randn(C0, generator=g, dtype=torch.float64) * 0.3)
        m = masks(nc0, [0, 0, 0, 2, 0, 0, 3, 4, 0])
        assert len(m) == 8 and m[7].shape == (H // 2, W // 2)
        cv0, cv1 = _layer(g, C0, C0, 3), _layer(g, C0, C0, 3)   # conv_0 / conv_1 behind the transposed convs (rpn_v1.py:200-210)
        # the dense chain
        a = _cbr(x, b0[0]); b = _cbr(a, b0[1]); x0 = _cbr(b, b0[2])
        ha = _cbr(x0, b1[0], 2); hb = _cbr(ha, b1[1]); x1 = _cbr(hb, b1[2])
        t0, t1 = _cbr(x0, tr0), _cbr(x1, tr1)
        mid0, mid1 = _dbr(t1, da) + t0, _dbr(t1, db)
        # the constants' chain
        c = torch.zeros(C0, dtype=torch.float64)
        cs = []
        for layer in b0 + b1:
            c = _const(layer, c)
            cs.append(c)
        ct0, ct1 = _const(tr0, cs[2]), _const(tr1, cs[5])
        cda, cdb = _dconst(da, ct1) + ct0[None], _dconst(db, ct1)
        def holds(y, tm, f, cval, what):
            out = ~_up(tm, f)   # pixels of the tiles nobody computes
            err = (y[0] - cval[:, None, None]).abs().amax(0)
            assert float(err[out].max() if out.any() else 0.0) < 1e-12, (trial, what, float(err[out].max()))
            return float(out.float().mean())
        skipped = [holds(a, m[0], 2, cs[0], "b0.0"), holds(b, m[1], 2, cs[1], "b0.1"), holds(x0, m[2], 2, cs[2], "b0.2"),
                   holds(ha, m[3], 2, cs[3], "b1.0"), holds(hb, m[4], 2, cs[4], "b1.1"), holds(x1, m[5], 2, cs[5], "b1.2"),
                   holds(t0, m[2], 2, ct0, "trans_0"), holds(t1, m[5], 2, ct1, "trans_1")]
        # the transposed convs: a 4x4 output block per tile, one constant per output parity class
        out = ~_up(m[6], 4)
        for y, cv, what in ((mid0, cda, "deconv_0 + trans_0"), (mid1, cdb, "deconv_1")):
            for py in (0, 1):
                for px in (0, 1):
                    err = (y[0, :, py::2, px::2] - cv[py * 2 + px][:, None, None]).abs().amax(0)
                    o = out[py::2, px::2]
                    assert float(err[o].max() if o.any() else 0.0) < 1e-12, (trial, what, py, px)
        # conv_0 / conv_1 over the transposed convs' outputs: outside the slot's 2x2 tiles the output is one constant per parity class
        def _pconst(layer, cpar):
            w, s_, t_ = layer
            return torch.stack([torch.relu(s_ * sum(w[:, :, ky, kx] @ cpar[((py + ky - 1) & 1) * 2 + ((px + kx - 1) & 1)]
                                                     for ky in range(3) for kx in range(3)) + t_) for py in (0, 1) for px in (0, 1)])
        out2 = ~_up(m[7], 2)
        for y, layer, cpar, what in ((_cbr(mid0, cv0), cv0, cda, "conv_0"), (_cbr(mid1, cv1), cv1, cdb, "conv_1")):
            cv = _pconst(layer, cpar)
            for py in (0, 1):
                for px in (0, 1):
                    err = (y[0, :, py::2, px::2] - cv[py * 2 + px][:, None, None]).abs().amax(0)
                    o = out2[py::2, px::2]
                    assert float(err[o].max() if o.any() else 0.0) < 1e-12, (trial, what, py, px)
        if trial == 2:
            assert float(out2.float().mean()) > 0.05
        if trial == 2:
            # the rule is not trivially "everything" (on a 96 x 112 map the border ring is a tenth of the tiles)
            assert skipped[0] > 0.9 and skipped[2] > 0.5 and skipped[5] > 0.1 and float(out.float().mean()) > 0.1, (skipped, float(out.float().mean()))


def test_the_engine_constants_are_this_chain():
    """sessd_hip.engine.active_tile_constants (float64 over the folded weights; what DensePlan.act_const rounds to float32) against
    the neck's own conv / BatchNorm(eval) / ReLU modules applied to constant maps, read away from the border; the transposed convs
    per output parity class. (fold_bn folds BatchNorm in float32 -- the scale / shift the kernels apply --, hence 2e-6, not 1e-12.)"""
    from det3d.models import build_detector
    from sessd_hip import configs, synth
    from sessd_hip.engine import active_tile_constants
    model = build_detector(configs.kitti_car_model(), train_cfg=None, test_cfg=configs.TEST_CFG)
    synth.init_synthetic_weights(model, 3)
    g = torch.Generator().manual_seed(9)
    for mod in model.neck.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):   # statistics that make the fold matter
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.2)
            mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
    neck = model.neck.double().eval()
    chain = active_tile_constants(neck)
    assert len(chain) == 10 and chain[8][0].shape == (4, 128) and chain[9][0].shape == (4, 128)
    S = 12
    def on_constant(seq, ci, bi, c, pad=False):
        x = c[None, :, None, None].expand(1, -1, S, S).contiguous()
        with torch.no_grad():
            if pad:
                x = seq[0](x)   # ZeroPad2d(1) in front of the unpadded first conv
            y = torch.relu(seq[bi](seq[ci](x)))
        return y
    b0, b1 = neck.bottom_up_block_0, neck.bottom_up_block_1
    c = torch.zeros(128, dtype=torch.float64)
    for k, (seq, ci, bi, pad) in enumerate(((b0, 1, 2, True), (b0, 4, 5, False), (b0, 7, 8, False), (b1, 0, 1, False), (b1, 3, 4, False),
                                            (b1, 6, 7, False))):
        y = on_constant(seq, ci, bi, c, pad)
        c = y[0, :, y.shape[2] // 2, y.shape[3] // 2].clone()
        assert torch.allclose(chain[k], c, rtol=2e-6, atol=2e-6), k
        # ... and the map IS constant away from the border (one pixel of border for a 3x3 layer)
        assert float((y[0, :, 1:-1, 1:-1] - c[:, None, None]).abs().max()) < 1e-12, k   # (the modules' own arithmetic: exact)
    t0 = on_constant(neck.trans_0, 0, 1, chain[2])[0, :, 3, 3]
    t1 = on_constant(neck.trans_1, 0, 1, chain[5])[0, :, 3, 3]
    assert torch.allclose(chain[6], t0, rtol=2e-6, atol=2e-6) and torch.allclose(chain[7], t1, rtol=2e-6, atol=2e-6)
    for which, blk in ((0, neck.deconv_block_0), (1, neck.deconv_block_1)):
        y = on_constant(blk, 0, 1, chain[7])   # (1, 128, 2S, 2S)
        for py in (0, 1):
            for px in (0, 1):
                v = y[0, :, 6 + py, 6 + px] + (chain[6] if which == 0 else 0.0)
                assert torch.allclose(chain[8][which][py * 2 + px], v, rtol=2e-6, atol=2e-6), (which, py, px)
    # conv_0 / conv_1 on a map that holds the transposed convs' parity-class constants: the modules' own arithmetic, away from the border
    for which, blk in ((0, neck.conv_0), (1, neck.conv_1)):
        x = torch.zeros(1, 128, 2 * S, 2 * S, dtype=torch.float64)
        for py in (0, 1):
            for px in (0, 1):
                x[0, :, py::2, px::2] = chain[8][which][py * 2 + px][:, None, None]
        with torch.no_grad():
            y = torch.relu(blk[1](blk[0](x)))
        for py in (0, 1):
            for px in (0, 1):
                assert torch.allclose(chain[9][which][py * 2 + px], y[0, :, 6 + py, 6 + px], rtol=2e-6, atol=2e-6), (which, py, px)


def test_the_gpu_tests_use_the_same_rule():
    """tests/test_dense_active_gpu.py holds the kernels' masks to ITS restatement of the rule (per image, flattened, on the 200 x 176
    map): the same masks as the function above on the same sites, for every step program the GPU tests run."""
    import test_dense_active_gpu as G
    rng = np.random.RandomState(11)
    idx = G._sites(5, 2, 400)
    for steps in (3, [0, 0, 0, 1, 0, 0], [0, 0, 0, 2, 0, 0], [2, 0], [0, 0, 0, 2, 0, 0, 3], [0, 1, 3], [2, 3], [0, 2, 3],
                  [0, 0, 0, 2, 0, 0, 3, 4, 0], [2, 3, 4, 0]):
        want = G._masks_numpy(idx, 2, steps)
        st = [0] * steps if isinstance(steps, int) else steps
        for b in range(2):
            nc0 = np.zeros((G.H, G.W), bool)
            s = idx[idx[:, 0] == b]
            nc0[s[:, 2], s[:, 3]] = True
            mine = masks(nc0, st)
            assert len(mine) == len(want[b])
            for l, (a, w) in enumerate(zip(mine, want[b])):
                assert np.array_equal(a.reshape(-1), w), (steps, b, l)
