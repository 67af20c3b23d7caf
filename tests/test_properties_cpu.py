"""Size- and pose-independent properties of the host-side geometry the data path is built on (no golden vectors needed):
symmetry of the collision test, rigid-motion invariance of point membership, invertibility of the recorded global
transformation, epoch coverage of the database sampler, farthest-point property of the thinning, text round trips."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _boxes(rng, n, spread=12.0):
    b = np.zeros((n, 7))
    b[:, :2] = rng.uniform(-spread, spread, (n, 2)); b[:, 2] = rng.uniform(-1.5, 0.5, n)
    b[:, 3] = rng.uniform(1.2, 2.2, n); b[:, 4] = rng.uniform(3.0, 5.0, n); b[:, 5] = rng.uniform(1.3, 2.0, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def test_collision_test_is_symmetric_and_translation_invariant():
    from det3d.core.bbox import box_np_ops
    from det3d.core.sampler import preprocess as prep
    rng = np.random.RandomState(0)
    for trial in range(20):
        a, b = _boxes(rng, 9), _boxes(rng, 7)
        ca = box_np_ops.center_to_corner_box2d(a[:, :2], a[:, 3:5], a[:, 6])
        cb = box_np_ops.center_to_corner_box2d(b[:, :2], b[:, 3:5], b[:, 6])
        ab = prep.box_collision_test(ca, cb)
        assert np.array_equal(ab, prep.box_collision_test(cb, ca).T)
        shift = rng.uniform(-30, 30, 2)
        assert np.array_equal(ab, prep.box_collision_test(ca + shift, cb + shift))
        # colliding footprints have positive rotated overlap and vice versa (the oracle's polygon clipper as referee)
        from oracle import capi
        for i in range(len(a)):
            for j in range(len(b)):
                iou = capi.quad_iou(ca[i].astype(np.float32), cb[j].astype(np.float32))
                if iou > 1e-4:
                    assert ab[i, j], (trial, i, j, iou)
                if not ab[i, j]:
                    assert iou < 1e-4
    far = ca + 1000.0
    assert not prep.box_collision_test(ca, far).any()


def test_point_membership_is_invariant_under_rigid_motion():
    from det3d.core.bbox import box_np_ops
    rng = np.random.RandomState(1)
    boxes = _boxes(rng, 6, spread=8.0)
    pts = np.concatenate([rng.uniform(-12, 12, (4000, 2)), rng.uniform(-2.5, 1.5, (4000, 1))], 1)
    m0 = box_np_ops.points_in_rbbox(pts, boxes)
    assert 50 < m0.sum() < 3000
    # distance of every point to the nearest box face, to leave out the points rounding could flip
    corners = box_np_ops.center_to_corner_box3d(boxes[:, :3], boxes[:, 3:6], boxes[:, 6], origin=(0.5, 0.5, 0.5), axis=2)
    from det3d.core.bbox.geometry import surface_equ_3d_jitv2
    n, d = surface_equ_3d_jitv2(box_np_ops.corner_to_surfaces_3d(corners)[:, :, :3, :])
    dist = np.abs((pts[:, None, None, :] * n[None]).sum(-1) + d[None]) / np.linalg.norm(n, axis=-1)[None]
    safe = dist.min(axis=2) > 1e-6
    for angle, shift in ((0.7, [3.0, -2.0, 0.4]), (-2.1, [-10.0, 5.0, -1.0])):
        p2 = box_np_ops.rotation_points_single_angle(pts.copy(), angle, axis=2) + shift
        b2 = boxes.copy()
        b2[:, :3] = box_np_ops.rotation_points_single_angle(b2[:, :3], angle, axis=2) + shift
        b2[:, 6] += angle
        m1 = box_np_ops.points_in_rbbox(p2, b2)
        assert np.array_equal(m0[safe], m1[safe])
    assert np.array_equal(box_np_ops.points_count_rbbox(pts, boxes), m0.sum(0))


def test_recorded_global_transformation_is_invertible():
    from det3d.core.sampler import preprocess as prep
    rng = np.random.RandomState(2)
    for seed in range(6):
        pts = rng.uniform(-40, 40, (500, 4)).astype(np.float32)
        boxes = _boxes(rng, 5).astype(np.float32)
        p, b = pts.copy(), boxes.copy()
        np.random.seed(seed)
        b, p, flipped = prep.random_flip_v2(b, p)
        b, p, rot = prep.global_rotation_v3(b, p, [-0.785, 0.785])
        b, p, scale = prep.global_scaling_v3(b, p, 0.95, 1.05)
        q = p[:, :3] / scale
        s, c = np.sin(-rot), np.cos(-rot)
        q = np.stack([q[:, 0] * c + q[:, 1] * s, -q[:, 0] * s + q[:, 1] * c, q[:, 2]], 1)
        if flipped:
            q[:, 1] = -q[:, 1]
        assert np.allclose(q, pts[:, :3], atol=2e-4) and np.array_equal(p[:, 3], pts[:, 3])
        assert np.allclose(b[:, 3:6] / scale, boxes[:, 3:6], atol=1e-5)


def test_batch_sampler_visits_everything_once_per_pass():
    from det3d.core.sampler.preprocess import BatchSampler
    np.random.seed(0)
    s = BatchSampler(list(range(23)), "x")
    seen = []
    while len(seen) < 23:
        got = s.sample(5)
        assert 0 < len(got) <= 5
        seen += got
    assert sorted(seen) == list(range(23))          # the short tail ends the pass; then it reshuffles
    nxt = s.sample(5)
    assert len(nxt) == 5 and len(set(nxt)) == 5


def test_farthest_point_sampling_property():
    from scipy.spatial import cKDTree
    from det3d.datasets.utils.sa_da_v2 import ifp_sample
    rng = np.random.RandomState(3)
    x = rng.uniform(-1, 1, (200, 3))
    d, i = cKDTree(x).query(x, x.shape[0])
    pick = ifp_sample(d, i, 40)
    assert pick[0] == 0 and len(set(pick.tolist())) == 40
    for k in range(1, 40):   # every pick maximises the distance to the points picked before it
        dist = np.linalg.norm(x[:, None, :] - x[pick[:k]][None], axis=-1).min(axis=1)
        assert abs(dist[pick[k]] - dist.max()) < 1e-12


def test_label_text_round_trip():
    from det3d.datasets.kitti import kitti_common as K
    rng = np.random.RandomState(4)
    n = 9
    a = dict(name=np.array(["Car", "Van", "Pedestrian"] * 3), truncated=rng.uniform(0, 1, n).round(2), occluded=rng.randint(0, 3, n),
             alpha=rng.uniform(-3, 3, n).round(2), bbox=rng.uniform(0, 1200, (n, 4)).round(2), dimensions=rng.uniform(1, 5, (n, 3)).round(2),
             location=rng.uniform(-40, 70, (n, 3)).round(2), rotation_y=rng.uniform(-3, 3, n).round(2), score=rng.uniform(0, 1, n).round(4))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "000001.txt")
        with open(path, "w") as f:
            f.write("\n".join(K.annos_to_kitti_label(a)))
        b = K.get_label_anno(path)
    for k in ("truncated", "alpha", "bbox", "dimensions", "location", "rotation_y", "score"):
        assert np.allclose(a[k], b[k], atol=1e-4), k
    assert list(a["name"]) == list(b["name"]) and np.array_equal(a["occluded"], b["occluded"])
    assert np.array_equal(b["index"], np.arange(n)) and np.array_equal(b["group_ids"], np.arange(n))
