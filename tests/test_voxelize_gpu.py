"""HIP voxelizer vs the reference golden vectors and the CPU oracle: BIT-EXACT (integer indices and
the float32 payload are copies of the input points; the mean is one float32 sum chain + one divide)."""
import os

import numpy as np
import pytest
import torch

import oracle
from sessd_hip import ops, synth

pytestmark = pytest.mark.gpu


def _run(points, mp, mv, dev, coors4=False):
    pts = torch.from_numpy(np.ascontiguousarray(points)).to(dev)
    r = ops.voxelize_batch([pts], synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, mv, with_batch_index=coors4)
    m = int(r["prefix"][1].item())
    return (r["voxels"][:m].cpu().numpy(), r["coors"][:m].cpu().numpy(), r["num_points"][:m].cpu().numpy(),
            r["mean"][:m].cpu().numpy())


@pytest.mark.parametrize("case", ["frame", "cap", "edge", "dup", "mp35", "empty"])
def test_golden_cases(golden_dir, dev, case):
    g = np.load(os.path.join(golden_dir, "voxelize_ref.npz"))
    mp, mv = [int(x) for x in g[case + "_cfg"]]
    v, c, n, mean = _run(g[case + "_pts"], mp, mv, dev)
    assert v.shape == g[case + "_voxels"].shape
    assert np.array_equal(c, g[case + "_coors"])
    assert np.array_equal(n, g[case + "_num"])
    assert np.array_equal(v.view(np.uint32), g[case + "_voxels"].view(np.uint32))
    if v.shape[0]:
        want = oracle.vfe_mean(g[case + "_voxels"], g[case + "_num"], 4)
        assert np.array_equal(mean.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("seed,npts,mv", [(0, 20000, 16000), (1, 20000, 20000), (2, None, 16000), (5, None, 64000)])
def test_full_size_vs_oracle(dev, seed, npts, mv):
    pts = synth.make_frame(seed, npts, supersample=1 if mv < 64000 else 3)
    v, c, n, mean = _run(pts, 5, mv, dev, coors4=True)
    ov, oc, on = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, mv)
    assert np.array_equal(c[:, 1:], oc) and (c[:, 0] == 0).all()
    assert np.array_equal(n, on)
    assert np.array_equal(v.view(np.uint32), ov.view(np.uint32))
    assert np.array_equal(mean.view(np.uint32), oracle.vfe_mean(ov, on, 4).view(np.uint32))


def test_batch_and_properties(dev):
    frames = [synth.make_frame(s, 12000) for s in (10, 11, 12)]
    r = ops.voxelize_batch([torch.from_numpy(f).to(dev) for f in frames], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    prefix = r["prefix"].cpu().numpy()
    coors = r["coors"].cpu().numpy()
    for b, f in enumerate(frames):
        ov, oc, on = oracle.points_to_voxel(f, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
        lo, hi = prefix[b], prefix[b + 1]
        assert hi - lo == oc.shape[0]
        assert (coors[lo:hi, 0] == b).all()
        assert np.array_equal(coors[lo:hi, 1:], oc)
        assert np.array_equal(r["voxels"][lo:hi].cpu().numpy().view(np.uint32), ov.view(np.uint32))
    # size-independent properties: coordinates unique per frame, counts within [1, max_points]
    key = coors[:prefix[-1]].astype(np.int64)
    lin = ((key[:, 0] * 40 + key[:, 1]) * 1600 + key[:, 2]) * 1408 + key[:, 3]
    assert np.unique(lin).size == lin.size
    n = r["num_points"][:prefix[-1]].cpu().numpy()
    assert n.min() >= 1 and n.max() <= 5


def test_vfe_standalone(dev):
    pts = synth.make_frame(4, 8000)
    ov, oc, on = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 20000)
    got = ops.vfe_mean(torch.from_numpy(ov).to(dev), torch.from_numpy(on).to(dev), 4).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), oracle.vfe_mean(ov, on, 4).view(np.uint32))


def test_all_frames_of_a_batch_in_four_launches(dev):
    """sessd_voxelize_frames (round 4: the frames of a batch no longer wait for each other -- 4 launches instead of 4 per frame)
    against the per-frame entry point on one shared hash: every output bit-identical, on ragged frames (padding rows), a frame
    that breaks at max_voxels followed by frames that do not, an empty frame, and the hash values (level-0 site index)."""
    MV = 5000
    frames = [synth.make_frame(31, 20000), synth.make_frame(32, 20000)[:2500].copy(), np.zeros((0, 4), np.float32),
              synth.make_frame(33, 20000)[::7].copy(), synth.make_frame(34, 20000), synth.make_frame(35, 9000)]
    pts = [torch.from_numpy(np.ascontiguousarray(f)).to(dev) for f in frames]
    a = ops.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, MV)
    b = ops.voxelize_frames(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, MV)
    pa, pb = a["prefix"].cpu().numpy(), b["prefix"].cpu().numpy()
    assert np.array_equal(pa, pb) and pa[1] - pa[0] == MV and pa[3] == pa[2] and 0 < pa[2] - pa[1] < MV   # capped, empty, uncapped
    m = int(pa[-1])
    for k in ("voxels", "coors", "num_points", "mean"):
        assert torch.equal(a[k][:m], b[k][:m]), k
    # the oracle, frame by frame
    for i, f in enumerate(frames):
        v, c, n = oracle.points_to_voxel(f, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, MV)
        lo, hi = int(pb[i]), int(pb[i + 1])
        assert hi - lo == c.shape[0]
        assert np.array_equal(b["coors"][lo:hi, 1:].cpu().numpy(), c) and np.all(b["coors"][lo:hi, 0].cpu().numpy() == i)
        assert np.array_equal(b["num_points"][lo:hi].cpu().numpy(), n)
        assert np.array_equal(b["voxels"][lo:hi].cpu().numpy().view(np.uint32), v.view(np.uint32))
    # the shared hash maps cell -> global row in both: same set of (key, value) pairs
    ka, va = a["hash"].keys.cpu().numpy(), a["hash"].vals.cpu().numpy()
    kb, vb = b["hash"].keys.cpu().numpy(), b["hash"].vals.cpu().numpy()
    live_a, live_b = ka != 0x7F7F7F7F, kb != 0x7F7F7F7F
    da = dict(zip(ka[live_a].tolist(), va[live_a].tolist()))
    db = dict(zip(kb[live_b].tolist(), vb[live_b].tolist()))
    assert da == db


def test_a_wall_across_the_x_axis_fills_one_offset_class_of_the_hash(dev):
    """The hash keeps the 8 cells of an aligned x-run in one bucket: slot = bucket * 8 + (x & 7), and a taken slot sends the key
    to the same offset of the next bucket. A plane x = const puts EVERY cell into one offset class (an eighth of the table);
    with more cells than that class holds the probe sequence must move on to the next offset -- not report a full table.
    6000 distinct cells, capacity 2 * 6000 -> 16384 slots -> 2048 per class."""
    rng = np.random.RandomState(3)
    ny, nz = 150, 40
    yy, zz = np.meshgrid(np.arange(ny), np.arange(nz), indexing="ij")
    pts = np.stack([np.full(ny * nz, 20.0 + 0.025), -3.0 + 0.05 * yy.ravel() + 0.025, -3.0 + 0.1 * zz.ravel() + 0.05,
                    rng.rand(ny * nz)], 1).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    v, c, n, mean = _run(pts, 5, 16000, dev, coors4=True)
    ov, oc, on = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    assert oc.shape[0] == ny * nz and len(np.unique(oc[:, 2])) == 1
    assert np.array_equal(c[:, 1:], oc) and np.array_equal(n, on)
    assert np.array_equal(v.view(np.uint32), ov.view(np.uint32))
