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
