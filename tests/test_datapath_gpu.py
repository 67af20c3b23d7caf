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


def _same_point_set(a, b, atol=3e-4):
    """equal as multisets of rows (the stage ends with a shuffle; rows compared after a lexicographic sort)"""
    if a.shape != b.shape:
        return False
    ka, kb = np.lexsort(np.round(a, 3).T[::-1]), np.lexsort(np.round(b, 3).T[::-1])
    return bool(np.allclose(a[ka], b[kb], rtol=0, atol=atol))


def test_preprocess_stage_with_the_cloud_on_the_device(dev, golden_dir):
    """det3d.datasets.pipelines.Preprocess given a CUDA point tensor: GT-AUG removal of covered points, per-object
    noise, the points_raw snapshot + global flip / rotation / scaling and the shuffle run on the device kernels with the host
    stage's random draws -- same seed => the reference run's boxes, names, transformation and point sets (tests/golden/
    datapath_ref.npz F*: the reference's Preprocess from source), and the host stage's on the same frames."""
    import tempfile
    from make_golden_datapath import SAMPLER_CFG, make_database, make_scene, train_cfg
    from det3d.builder import build_dbsampler
    from det3d.datasets.pipelines import Preprocess
    G = np.load(os.path.join(golden_dir, "datapath_ref.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        db = make_database(tmp)
        out = {}
        for mode in ("device", "host"):
            cfg = train_cfg(); cfg["db_sampler"] = dict(SAMPLER_CFG)
            np.random.seed(500)
            stage = Preprocess(cfg=cfg, db_sampler=build_dbsampler(cfg["db_sampler"], db_infos=db))
            frames = []
            for k in range(2):
                p, b, n = make_scene(30 + k)
                pts = torch.from_numpy(p).to(dev) if mode == "device" else p
                res, _ = stage(dict(labeled=True, metadata=dict(image_prefix=tmp, num_point_features=4),
                                    lidar=dict(points=pts, annotations=dict(boxes=b, names=n))), None)
                frames.append(res)
            p, b, n = make_scene(40)
            pts = torch.from_numpy(p).to(dev) if mode == "device" else p
            res, _ = stage(dict(labeled=False, metadata=dict(image_prefix=tmp, num_point_features=4), lidar=dict(points=pts)), None)
            frames.append(res)
            out[mode] = frames
    host = lambda t: t.cpu().numpy() if torch.is_tensor(t) else t
    for k in range(2):
        d, h = out["device"][k]["lidar"], out["host"][k]["lidar"]
        assert d["points"].is_cuda and d["points_raw"].is_cuda
        assert list(d["annotations"]["gt_names"]) == list(G["F%d_names" % k]) == list(h["annotations"]["gt_names"])
        assert np.array_equal(d["annotations"]["gt_classes"], G["F%d_classes" % k])
        t = d["transformation"]
        assert np.allclose([float(t["flipped"]), t["noise_rotation"], t["noise_scale"]], G["F%d_t" % k], rtol=0, atol=1e-12)
        assert np.allclose(d["annotations"]["gt_boxes"], G["F%d_boxes" % k], rtol=0, atol=2e-5)
        assert np.allclose(d["annotations_raw"]["gt_boxes"], G["F%d_boxes_raw" % k], rtol=0, atol=2e-5)
        assert np.array_equal(d["annotations"]["gt_boxes"], h["annotations"]["gt_boxes"])
        assert _same_point_set(host(d["points_raw"]), G["F%d_points_raw" % k]) and _same_point_set(host(d["points"]), G["F%d_points" % k])
        assert _same_point_set(host(d["points_raw"]), h["points_raw"], atol=1e-5) and host(d["points"]).shape == h["points"].shape
        # same shuffle: row order identical to the host stage
        assert np.allclose(host(d["points"]), h["points"], rtol=0, atol=3e-4)
    d, h = out["device"][2]["lidar"], out["host"][2]["lidar"]
    assert "annotations" not in d and _same_point_set(host(d["points"]), G["F_unlabeled_points"])
    assert np.allclose(host(d["points"]), h["points"], rtol=0, atol=3e-4)
    t = d["transformation"]
    assert np.allclose([float(t["flipped"]), t["noise_rotation"], t["noise_scale"]], G["F_unlabeled_t"], rtol=0, atol=1e-12)
    # the next stage takes the device cloud as it is: voxels of the student's and the teacher's view without a host copy
    from det3d.datasets.pipelines import Voxelization
    vcfg = dict(range=[0, -40.0, -3.0, 70.4, 40.0, 1.0], voxel_size=[0.05, 0.05, 0.1], max_points_in_voxel=5, max_voxel_num=20000)
    rd, _ = Voxelization(cfg=vcfg)(out["device"][0], None)
    rh, _ = Voxelization(cfg=vcfg)(out["host"][0], None)
    for key in ("voxels", "voxels_raw"):
        vd, vh = rd["lidar"][key], rh["lidar"][key]
        assert torch.is_tensor(vd["voxels"]) and vd["voxels"].is_cuda
        assert abs(int(vd["voxels"].shape[0]) - int(vh["voxels"].shape[0])) <= max(2, int(vh["voxels"].shape[0]) // 200)
    # validation mode passes the device tensor through
    val = Preprocess(cfg=dict(mode="val", shuffle_points=False, remove_environment=False, remove_unknown_examples=False))
    res, _ = val(dict(labeled=False, lidar=dict(points=torch.from_numpy(p).to(dev))), None)
    assert res["mode"] == "val" and res["lidar"]["points"].is_cuda and np.array_equal(res["lidar"]["points"].cpu().numpy(), p)


def test_farthest_point_sampling_equals_the_host_restatement(dev):
    from scipy.spatial import cKDTree
    from det3d.datasets.utils import sa_da_v2
    from sessd_hip import ops
    rng = np.random.RandomState(4)
    for n, k in ((300, 50), (51, 50), (1, 1), (2000, 64), (97, 97)):
        p = np.concatenate([rng.uniform(-2, 2, (n, 3)), rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
        if n > 20:
            p[7] = p[3]                     # duplicate points: distance ties
        d, idx = cKDTree(p[:, :3]).query(p[:, :3], n)
        want = sa_da_v2.ifp_sample(d.reshape(n, -1), idx.reshape(n, -1), k)
        got = ops.farthest_point_sample(torch.from_numpy(p).to(dev), k).cpu().numpy()
        assert got.tolist() == want.tolist(), (n, k)
    with pytest.raises(Exception):
        ops.farthest_point_sample(torch.zeros((5000, 4), device=dev), 10)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_shape_aware_augmentation_on_the_device(dev, seed):
    """sa_da_v2.pyramid_augment_v0_device vs the host function with the same seed and high stage probabilities: the same rows in
    the same order (rest, then the thinned / swapped groups), coordinates to float32 rounding."""
    from make_golden_datapath import make_scene
    from det3d.datasets.utils import sa_da_v2
    pts, boxes, names = make_scene(20 + seed)
    kw = dict(enable_sa_dropout=0.3, enable_sa_sparsity=[0.6, 20], enable_sa_swap=[0.8, 10])
    np.random.seed(900 + seed)
    want = sa_da_v2.pyramid_augment_v0(boxes.copy(), pts.copy(), **kw)
    after_host = np.random.uniform()
    np.random.seed(900 + seed)
    got = sa_da_v2.pyramid_augment_v0_device(boxes.copy(), torch.from_numpy(pts.copy()).to(dev), **kw)
    after_dev = np.random.uniform()
    assert got.is_cuda and got.dtype == torch.float32
    assert after_host == after_dev                                   # the same number of random draws
    g = got.cpu().numpy()
    assert g.shape == want.shape and want.shape[0] < pts.shape[0]     # something was dropped / thinned
    assert np.allclose(g, want, rtol=0, atol=2e-4)
    assert float(np.abs(want[-200:, :3] - pts[-200:, :3]).max()) > 0  # the tail holds moved / re-ordered groups
