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
