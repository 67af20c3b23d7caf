"""sessd_points_in_bodies (device membership test of the training data path).
The masks must be IDENTICAL to the reference's numba loop: golden A_in_rbbox of tests/golden/datapath_ref.npz (the reference run
from source) and the host mirror on boxes, pyramids and a body count that needs more than one mask word."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

pytestmark = pytest.mark.gpu


def _planes(surfaces):
    from det3d.core.bbox.geometry import surface_equ_3d_jitv2
    n, d = surface_equ_3d_jitv2(surfaces[:, :, :3, :])
    return np.ascontiguousarray(np.concatenate([n, d[..., None]], axis=-1).astype(np.float32))


def test_masks_identical_to_the_reference_loop(dev, golden_dir):
    from make_golden_datapath import make_scene
    from det3d.core.bbox import box_np_ops
    from det3d.core.bbox.geometry import points_in_convex_polygon_3d_jit
    from det3d.datasets.utils import sa_da_v2
    from sessd_hip import ops
    G = np.load(os.path.join(golden_dir, "datapath_ref.npz"))
    pts, boxes, names = make_scene(1)
    d_pts = torch.from_numpy(pts).to(dev)
    corners = box_np_ops.center_to_corner_box3d(boxes[:, :3], boxes[:, 3:6], boxes[:, 6], origin=(0.5, 0.5, 0.5), axis=2)
    surf = box_np_ops.corner_to_surfaces_3d(corners)
    got = ops.points_in_bodies(d_pts, torch.from_numpy(_planes(surf)).to(dev)).cpu().numpy()
    assert np.array_equal(np.packbits(got), G["A_in_rbbox"])
    # pyramids: 5 faces, 42 bodies -> two mask words
    pyr = sa_da_v2.get_pyramids(boxes).reshape(-1, 15)
    v = pyr.reshape(-1, 5, 3)
    psurf = v[:, sa_da_v2._PYRAMID_FACES].reshape(-1, 5, 3, 3)
    got = ops.points_in_bodies(d_pts, torch.from_numpy(_planes(psurf)).to(dev)).cpu().numpy()
    assert got.shape == (pts.shape[0], 42) and np.array_equal(got, points_in_convex_polygon_3d_jit(pts[:, :3], psurf))
    # ragged sizes: 1 point, 1 body; 0 points; 97 bodies (4 words)
    one = ops.points_in_bodies(d_pts[:1], torch.from_numpy(_planes(surf[:1])).to(dev)).cpu().numpy()
    assert np.array_equal(one, points_in_convex_polygon_3d_jit(pts[:1, :3], surf[:1]))
    assert ops.points_in_bodies(d_pts[:0], torch.from_numpy(_planes(surf)).to(dev)).shape == (0, len(boxes))
    rng = np.random.RandomState(0)
    many = np.repeat(boxes, 14, axis=0)[:97].copy()
    many[:, :2] += rng.uniform(-2, 2, (97, 2)).astype(np.float32)
    msurf = box_np_ops.corner_to_surfaces_3d(box_np_ops.center_to_corner_box3d(many[:, :3], many[:, 3:6], many[:, 6], origin=(0.5, 0.5, 0.5), axis=2))
    got = ops.points_in_bodies(d_pts, torch.from_numpy(_planes(msurf)).to(dev)).cpu().numpy()
    assert np.array_equal(got, points_in_convex_polygon_3d_jit(pts[:, :3], msurf)) and got.any(0).sum() > 50


def _box_planes(boxes, dev, extra=(0.0, 0.0, 0.0)):
    from det3d.core.bbox import box_np_ops
    corners = box_np_ops.center_to_corner_box3d(boxes[:, :3], boxes[:, 3:6] + np.asarray(extra, boxes.dtype), boxes[:, 6],
                                                origin=[0.5, 0.5, 0.5], axis=2)
    return torch.from_numpy(_planes(box_np_ops.corner_to_surfaces_3d_jit(corners))).to(dev)


def test_rigid_moves_equal_the_reference_run(dev, golden_dir):
    """noise_per_object_v4_ with the point side on the device: the host draws the noise and resolves the collisions (box level,
    same RNG order as the reference), sessd_points_rigid_moves moves the points. Golden B0 / B1: the reference run from source."""
    from make_golden_datapath import make_scene
    from det3d.core.sampler import preprocess as prep
    from sessd_hip import ops
    G = np.load(os.path.join(golden_dir, "datapath_ref.npz"))
    pts, boxes, names = make_scene(1)
    valid = np.array([n in ("Car", "Van") for n in names])
    for seed in (0, 1):
        b = boxes.copy()
        np.random.seed(100 + seed)
        n = b.shape[0]
        loc_noises = np.random.normal(scale=np.array([1.0, 1.0, 0.5], b.dtype), size=[n, 100, 3])
        rot_noises = np.random.uniform(-0.785, 0.785, size=[n, 100])
        chosen = prep.noise_per_box(b[:, [0, 1, 3, 4, 6]] + [0.0, 0.0, 0.0, 0.0, 0.0], valid, loc_noises, rot_noises)
        loc_t, rot_t = prep._select_transform(loc_noises, chosen), prep._select_transform(rot_noises, chosen)
        d_pts = torch.from_numpy(pts.copy()).to(dev)
        ops.points_rigid_moves_(d_pts, _box_planes(b, dev), b[:, :3], loc_t, rot_t, valid)
        got = d_pts.cpu().numpy()
        want = G["B%d_points" % seed]
        assert np.abs(got - want).max() <= 2e-5, seed
        assert (np.abs(got - pts).max(axis=1) > 1e-3).sum() > 1500   # the points inside the moved boxes moved with them
        assert np.array_equal(got[:, 3], pts[:, 3])                  # intensity untouched
        # and bit-identical to the host mirror (same float32 steps)
        host = pts.copy()
        masks = ops.points_in_bodies(torch.from_numpy(pts).to(dev), _box_planes(b, dev)).cpu().numpy()
        prep.points_transform_(host, b[:, :3], masks, loc_t, rot_t, valid)
        assert np.array_equal(got, host)


def test_global_transform_equals_the_reference_run(dev, golden_dir):
    from make_golden_datapath import make_scene
    from sessd_hip import ops
    G = np.load(os.path.join(golden_dir, "datapath_ref.npz"))
    pts, boxes, names = make_scene(1)
    flips = []
    for seed in range(4):
        f, r, s = G["C%d_t" % seed]
        d_pts = torch.from_numpy(pts.copy()).to(dev)
        raw = torch.empty_like(d_pts)
        ops.points_global_transform_(d_pts, bool(f), float(r), float(s), raw_copy=raw)
        assert np.abs(d_pts.cpu().numpy() - G["C%d_points" % seed]).max() <= 2e-5 * 80, seed   # coordinates up to ~80 m, float32
        assert np.array_equal(raw.cpu().numpy(), pts) and np.array_equal(d_pts[:, 3].cpu().numpy(), pts[:, 3])
        flips.append(bool(f))
    assert any(flips) and not all(flips)
    # empty cloud, identity transform
    ops.points_global_transform_(torch.empty((0, 4), device=dev), False, 0.0, 1.0)
    same = torch.from_numpy(pts.copy()).to(dev)
    ops.points_global_transform_(same, False, 0.0, 1.0)
    assert np.array_equal(same.cpu().numpy(), pts)


def test_compaction_feeds_the_voxelizer(dev):
    """Removing the points covered by pasted boxes (pipelines/preprocess.py:102-105) entirely on the device: membership ->
    keep flags -> ordered compaction -> voxelizer; identical to the host path point for point and voxel for voxel."""
    from make_golden_datapath import make_scene
    from det3d.core.bbox import box_np_ops
    from oracle import capi
    from sessd_hip import ops, synth
    pts, boxes, names = make_scene(1)
    d_pts = torch.from_numpy(pts).to(dev)
    inside = ops.points_in_bodies(d_pts, _box_planes(boxes[:3], dev))
    keep = ~inside.any(-1)
    out, n_out = ops.points_compact(d_pts, keep)
    n = int(n_out.item())
    want = pts[~box_np_ops.points_in_rbbox(pts, boxes[:3]).any(-1)]
    assert n == want.shape[0] and 0 < n < pts.shape[0]
    assert np.array_equal(out[:n].cpu().numpy(), want)
    r = ops.voxelize_batch([out[:n]], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    m = int(r["prefix"][1].item())
    v, c, k = capi.points_to_voxel(want, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    assert m == c.shape[0] and np.array_equal(r["coors"][:m, 1:].cpu().numpy(), c) and np.array_equal(r["voxels"][:m].cpu().numpy(), v)
    # ragged: nothing kept / everything kept / a length that is not a multiple of the block size
    e, ne = ops.points_compact(d_pts, torch.zeros(pts.shape[0], dtype=torch.bool, device=dev))
    a, na = ops.points_compact(d_pts[:1001], torch.ones(1001, dtype=torch.bool, device=dev))
    assert int(ne.item()) == 0 and int(na.item()) == 1001 and np.array_equal(a[:1001].cpu().numpy(), pts[:1001])
