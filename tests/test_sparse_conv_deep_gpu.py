"""sessd_sparse_conv_deep (four-deep operand ring, csrc/sparse_conv_deep.hip) vs sessd_sparse_conv: same packed weights, same
rulebook, results must be BIT-IDENTICAL (same accumulation order), for every channel pair of SpMiddleFHD, every cout split, partial
last tiles, empty offset masks, the dense BEV output, and the whole engine.

EXPERIMENTAL: written after round 1's GPU budget was spent, not yet run on hardware -> runs only with SESSD_EXPERIMENTAL=1
(    SESSD_EXPERIMENTAL=1 python -m pytest tests/test_sparse_conv_deep_gpu.py -x -q )."""
import os

import numpy as np
import pytest
import torch

from sessd_hip import ops

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("SESSD_EXPERIMENTAL") != "1",
                                                  reason="not yet validated on hardware; set SESSD_EXPERIMENTAL=1")]


def _sites(rng, B, shape, n):
    cells = B * shape[0] * shape[1] * shape[2]
    lin = rng.permutation(cells)[:n]
    x = lin % shape[2]; y = (lin // shape[2]) % shape[1]; z = (lin // (shape[2] * shape[1])) % shape[0]; b = lin // (shape[2] * shape[1] * shape[0])
    return np.stack([b, z, y, x], 1).astype(np.int32)


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)])
@pytest.mark.parametrize("n,cap", [(1500, 2048), (1501, 1501), (7, 64), (16, 16)])
def test_bit_identical_to_the_shipped_kernel(dev, cin, cout, n, cap):
    rng = np.random.RandomState(cin + cout + n)
    B, shape = 2, [9, 24, 20]
    idx = _sites(rng, B, shape, n)
    d_idx = torch.zeros((cap, 4), dtype=torch.int32, device=dev); d_idx[:n] = torch.from_numpy(idx).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    feat = torch.zeros((cap, cin), device=dev); feat[:n] = torch.randn(n, cin, generator=torch.Generator().manual_seed(1)).to(dev)
    w = (torch.randn(3, 3, 3, cin, cout, generator=torch.Generator().manual_seed(2)) * 0.2).to(dev)
    wpk = ops.sparse_pack_weight(w)
    scale, shift = (torch.rand(cout) + 0.5).to(dev), (torch.randn(cout) * 0.1).to(dev)
    h = ops.sparse_hash_build(d_idx, n_dev, shape)
    nbr, tm = ops.sparse_rulebook(d_idx, n_dev, 3, 1, 1, h)
    for split in (0, 1, 2, 4):
        if split > 1 and (cout // 16) % split:
            continue
        a = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=split)
        b = ops.sparse_conv(feat, nbr, tm, n_dev, wpk, cin, cout, scale, shift, True, cout_split=split, deep=True)
        torch.cuda.synchronize()
        assert torch.equal(a[:n], b[:n]), (cin, cout, split)
        assert float(a[:n].abs().max()) > 0
    # a strided conv's rulebook (sparser masks, some tiles with a single offset) and the dense BEV output
    out_idx, n_out, out_hash, err = ops.sparse_downsample_sites(d_idx, n_dev, 3, 2, 1, [5, 12, 10], 4096)
    nbr2, tm2 = ops.sparse_rulebook(out_idx, n_out, 3, 2, 1, h)
    a = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False)
    b = ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, False, deep=True)
    m = int(n_out.item())
    assert m > 0 and torch.equal(a[:m], b[:m])
    dense_a = torch.zeros((B, cout * 5, 12, 10), device=dev); dense_b = torch.zeros_like(dense_a)
    ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, True, dense_out=dense_a, out_indices=out_idx, dense_dims=[5, 12, 10])
    ops.sparse_conv(feat, nbr2, tm2, n_out, wpk, cin, cout, scale, shift, True, dense_out=dense_b, out_indices=out_idx, dense_dims=[5, 12, 10], deep=True)
    assert torch.equal(dense_a, dense_b) and float(dense_a.abs().max()) > 0


def test_engine_bit_identical_and_timing(dev):
    from sessd_hip import configs, synth
    from sessd_hip.engine import InferenceEngine
    VG = configs.VOXEL_GENERATOR
    model = configs.build_synthetic_detector(dev, seed=0, max_voxels=16000, num_points=20000)
    frame = [torch.from_numpy(synth.make_frame(3, 20000)).to(dev)]
    outs = []
    for deep in (False, True):
        e = InferenceEngine(model, VG["range"], VG["voxel_size"], VG["max_points_in_voxel"], 16000, configs.TEST_CFG, batch_size=1,
                            max_points_per_frame=20000, device=dev, deep_sparse=deep)
        e.set_points(frame); e.enqueue(); torch.cuda.synchronize()
        outs.append((e.results(), e.bev.clone(), e.stage_times(reps=10)["spmiddle"]))
    (r0, bev0, t0), (r1, bev1, t1) = outs
    assert torch.equal(bev0, bev1) and all(np.array_equal(r0[0][k], r1[0][k]) for k in r0[0])
    print("spmiddle eager ms: shipped %.3f, deep %.3f" % (t0, t1))
