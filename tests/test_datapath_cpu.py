"""The training data path upstream of the voxelizer (host stage as in the reference): the det3d mirror vs
tests/golden/datapath_ref.npz = the reference's own functions run from source on the same synthetic frames, database and seeds
(tests/golden/make_golden_datapath.py). Decisions (masks, chosen noise candidates, which objects are pasted, which points
survive) must be identical; coordinates agree to float32 rounding (the reference rotates with tiny matmuls, the mirror with the
expanded 2-term sums)."""
import os
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

TOL = dict(rtol=0, atol=2e-5)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "datapath_ref.npz"))


@pytest.fixture(scope="module")
def scene():
    from make_golden_datapath import make_scene
    pts, boxes, names = make_scene(1)
    return pts, boxes, names, np.array([n in ("Car", "Van") for n in names])


def test_point_and_box_predicates(G, scene):
    from make_golden_datapath import containment_case
    from det3d.core.bbox import box_np_ops
    from det3d.core.sampler import preprocess as prep
    pts, boxes, names, valid = scene
    m = box_np_ops.points_in_rbbox(pts, boxes)
    assert np.array_equal(np.packbits(m), G["A_in_rbbox"]) and m.sum() > 2000
    assert np.array_equal(box_np_ops.points_count_rbbox(pts, boxes), m.sum(0))
    quads = containment_case()
    hit = prep.box_collision_test(quads, quads)
    assert np.array_equal(hit, G["A_collision"])
    # crossing edges (0,2); strict containment of 1 in 0 both ways round; bounding rectangles touch only (0,3); far away (4)
    assert hit[0, 2] and hit[0, 1] and hit[1, 0] and not hit[0, 3] and not hit[4].any()   # (a box does not "collide" with its exact copy)
    corners = box_np_ops.center_to_corner_box2d(boxes[:, :2], boxes[:, 3:5], boxes[:, 6])
    assert np.array_equal(prep.box_collision_test(corners, corners), G["A_collision_scene"])
    edge = boxes.copy(); edge[0, :2] = [70.2, 39.9]; edge[1, :2] = [72.9, 0.0]; edge[2, :2] = [-2.6, 10.0]
    rng_ = np.array([0, -40.0, 70.4, 40.0], np.float32)
    assert np.array_equal(prep.filter_gt_box_outside_range(edge, rng_), G["A_range_mask"])
    assert np.array_equal(prep.filter_gt_box_outside_range_by_center(edge, rng_), G["A_center_mask"])
    assert not G["A_range_mask"].all() and G["A_range_mask"].sum() > G["A_center_mask"].sum() - 1


def test_noise_candidates_and_per_object_noise(G, scene):
    from det3d.core.sampler import preprocess as prep
    pts, boxes, names, valid = scene
    chosen = prep.noise_per_box(G["A_noise_boxes"].copy(), G["A_noise_valid"], G["A_noise_loc"], G["A_noise_rot"])
    assert np.array_equal(chosen, G["A_noise_chosen"])
    assert (chosen[~G["A_noise_valid"]] == -1).all() and (chosen[G["A_noise_valid"]] > 0).any()   # some first draws collide
    for seed in (0, 1):
        p, b = pts.copy(), boxes.copy()
        np.random.seed(100 + seed)
        prep.noise_per_object_v4_(b, p, valid, rotation_perturb=[-0.785, 0.785], center_noise_std=[1.0, 1.0, 0.5],
                                  global_random_rot_range=[0, 0], group_ids=None, num_try=100, data_aug_with_context=-1.0,
                                  data_aug_random_drop=-1.0)
        assert np.allclose(b, G["B%d_boxes" % seed], **TOL) and np.allclose(p, G["B%d_points" % seed], **TOL)
        assert np.array_equal(b[~valid], boxes[~valid]) and not np.allclose(b[valid, :2], boxes[valid, :2], atol=1e-3)
        assert (np.abs(p - pts).max(axis=1) > 1e-3).sum() > 1500    # the points inside the moved boxes moved with them


def test_global_transformations(G, scene):
    from det3d.core.sampler import preprocess as prep
    pts, boxes, names, valid = scene
    flips = []
    for seed in range(4):
        p, b = pts.copy(), boxes.copy()
        np.random.seed(200 + seed)
        b, p, f = prep.random_flip_v2(b, p)
        b, p, r = prep.global_rotation_v3(b, p, [-0.785, 0.785])
        b, p, s = prep.global_scaling_v3(b, p, 0.95, 1.05)
        assert np.allclose([float(f), r, s], G["C%d_t" % seed], rtol=0, atol=1e-12)
        assert np.allclose(b, G["C%d_boxes" % seed], **TOL) and np.allclose(p, G["C%d_points" % seed], **TOL)
        flips.append(bool(f))
    assert any(flips) and not all(flips)
    # unlabeled frames: no boxes
    p = pts.copy()
    np.random.seed(5)
    none, p, f = prep.random_flip_v2(None, p)
    none, p, r = prep.global_rotation_v3(None, p, 0.3)
    none, p, s = prep.global_scaling_v3(None, p)
    assert none is None and abs(r) <= 0.3 and 0.95 <= s <= 1.05


def _same_point_set(a, b, atol=2e-5):
    """equal as ordered arrays (the augmentations keep a deterministic order)"""
    return a.shape == b.shape and a.dtype == b.dtype and np.allclose(a, b, rtol=0, atol=atol)


def test_shape_aware_augmentation(G, scene):
    from det3d.datasets.utils import sa_da_v2
    pts, boxes, names, valid = scene
    cars = boxes[valid]
    assert np.allclose(sa_da_v2.get_pyramids(cars), G["D_pyramids"], rtol=0, atol=1e-6)
    for tag, kw, seed in (("drop", dict(enable_sa_dropout=0.6, enable_sa_sparsity=None, enable_sa_swap=None), 300),
                          ("sparse", dict(enable_sa_dropout=None, enable_sa_sparsity=[0.7, 30], enable_sa_swap=None), 301),
                          ("swap", dict(enable_sa_dropout=None, enable_sa_sparsity=None, enable_sa_swap=[0.7, 20]), 302),
                          ("mix", dict(enable_sa_dropout=0.25, enable_sa_sparsity=[0.05, 50], enable_sa_swap=[0.1, 50]), 303),
                          ("mix2", dict(enable_sa_dropout=0.4, enable_sa_sparsity=[0.4, 30], enable_sa_swap=[0.5, 20]), 304)):
        np.random.seed(seed)
        out = sa_da_v2.pyramid_augment_v0(cars.copy(), pts.copy(), **kw)
        assert _same_point_set(out, G["D_" + tag]), tag
    # what the stages do: dropout / thinning remove points, the swap moves them
    assert G["D_drop"].shape[0] < pts.shape[0] and G["D_sparse"].shape[0] < pts.shape[0]
    assert G["D_swap"].shape[0] == pts.shape[0] and not np.allclose(np.sort(G["D_swap"][:, 0]), np.sort(pts[:, 0]), atol=1e-4)
    # the six pyramids of a box tile it: every point inside a box is in at least one pyramid, points outside in none
    from det3d.core.bbox import box_np_ops
    inside = box_np_ops.points_in_rbbox(pts, cars).any(1)
    in_pyr = sa_da_v2.points_in_pyramids_mask(pts, sa_da_v2.get_pyramids(cars).reshape(-1, 15)).any(1)
    assert not in_pyr[~inside].any() and in_pyr[inside].mean() > 0.99   # points exactly on a shared face belong to no pyramid


def test_farthest_point_restatement():
    """ifp_sample (restated; the external package is absent): classic farthest-point order on a line of points."""
    from scipy.spatial import cKDTree
    from det3d.datasets.utils.sa_da_v2 import ifp_sample
    x = np.array([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [10.0, 0, 0], [4.9, 0, 0]])
    d, i = cKDTree(x).query(x, x.shape[0])
    assert list(ifp_sample(d, i, 4)) == [0, 3, 4, 2]
    with pytest.raises(ValueError):
        ifp_sample(d, i, 6)


def test_ground_truth_database_sampler(G):
    from make_golden_datapath import SAMPLER_CFG, make_database, make_scene
    from det3d.builder import build_dbsampler
    with tempfile.TemporaryDirectory() as tmp:
        db = make_database(tmp)
        np.random.seed(400)
        sampler = build_dbsampler(SAMPLER_CFG, db_infos=db)
        assert all(i["num_points_in_gt"] >= 5 and i["difficulty"] != -1 for i in sampler.db_infos["Car"])
        for k in range(3):
            _, b, n = make_scene(20 + k)
            got = sampler.sample_all(tmp, b, n, 4, False, gt_group_ids=None, calib=None, targeted_class_names=["Car", "Van"])
            assert list(got["gt_names"]) == list(G["E%d_names" % k])
            assert np.array_equal(got["gt_boxes"], G["E%d_boxes" % k]) and np.array_equal(got["points"], G["E%d_points" % k])
            assert got["gt_masks"].all() and list(got["group_ids"]) == list(range(len(b), len(b) + len(got["gt_names"])))
            # pasted objects collide neither with the frame's boxes nor with each other
            from det3d.core.bbox import box_np_ops
            from det3d.core.sampler import preprocess as prep
            allb = np.concatenate([b, got["gt_boxes"]])
            c = box_np_ops.center_to_corner_box2d(allb[:, :2], allb[:, 3:5], allb[:, 6])
            hit = prep.box_collision_test(c, c)
            np.fill_diagonal(hit, False)
            assert not hit.any()


def test_preprocess_stage_matches_reference_run(G):
    from make_golden_datapath import SAMPLER_CFG, make_database, make_scene, train_cfg
    from det3d.datasets.pipelines import Preprocess
    with tempfile.TemporaryDirectory() as tmp:
        db = make_database(tmp)
        cfg = train_cfg(); cfg["db_sampler"] = dict(SAMPLER_CFG)
        np.random.seed(500)
        from det3d.builder import build_dbsampler
        stage = Preprocess(cfg=cfg, db_sampler=build_dbsampler(cfg["db_sampler"], db_infos=db))
        assert cfg["class_names"] == ["Car", "Van"]   # enable_similar_type widens the configured list, like the reference
        for k in range(2):
            p, b, n = make_scene(30 + k)
            res = dict(labeled=True, metadata=dict(image_prefix=tmp, num_point_features=4),
                       lidar=dict(points=p, annotations=dict(boxes=b, names=n)))
            res, _ = stage(res, None)
            L = res["lidar"]
            assert res["mode"] == "train" and list(L["annotations"]["gt_names"]) == list(G["F%d_names" % k])
            assert np.array_equal(L["annotations"]["gt_classes"], G["F%d_classes" % k]) and L["annotations"]["gt_classes"].dtype == np.int32
            t = L["transformation"]
            assert np.allclose([float(t["flipped"]), t["noise_rotation"], t["noise_scale"]], G["F%d_t" % k], rtol=0, atol=1e-12)
            assert np.allclose(L["annotations"]["gt_boxes"], G["F%d_boxes" % k], **TOL)
            assert np.allclose(L["annotations_raw"]["gt_boxes"], G["F%d_boxes_raw" % k], **TOL)
            assert _same_point_set(L["points_raw"], G["F%d_points_raw" % k]) and _same_point_set(L["points"], G["F%d_points" % k])
            assert list(L["annotations_raw"]["gt_names"]) == list(L["annotations"]["gt_names"]) and "Pedestrian" not in L["annotations"]["gt_names"]
            # the recorded transformation maps the teacher's (raw) boxes onto the student's: what consistency_loss relies on
            raw, cur = L["annotations_raw"]["gt_boxes"].copy(), L["annotations"]["gt_boxes"]
            if t["flipped"]:
                raw[:, 1] = -raw[:, 1]; raw[:, 6] = -raw[:, 6] + np.pi
            s_, c_ = np.sin(t["noise_rotation"]), np.cos(t["noise_rotation"])
            x, y = raw[:, 0].copy(), raw[:, 1].copy()
            raw[:, 0], raw[:, 1] = x * c_ + y * s_, -x * s_ + y * c_
            raw[:, 6] += t["noise_rotation"]
            raw[:, :6] *= t["noise_scale"]
            assert np.allclose(raw, cur, rtol=0, atol=1e-4)
        p, b, n = make_scene(40)
        res, _ = stage(dict(labeled=False, metadata=dict(image_prefix=tmp, num_point_features=4), lidar=dict(points=p)), None)
        t = res["lidar"]["transformation"]
        assert _same_point_set(res["lidar"]["points"], G["F_unlabeled_points"]) and "annotations" not in res["lidar"]
        assert np.allclose([float(t["flipped"]), t["noise_rotation"], t["noise_scale"]], G["F_unlabeled_t"], rtol=0, atol=1e-12)
    val = Preprocess(cfg=dict(mode="val", shuffle_points=False, remove_environment=False, remove_unknown_examples=False))
    res, _ = val(dict(labeled=False, lidar=dict(points=p.copy())), None)
    assert res["mode"] == "val" and np.array_equal(res["lidar"]["points"], p)
    with pytest.raises(NotImplementedError):
        Preprocess(cfg=dict(mode="val", shuffle_points=False, remove_environment=True))


def test_loading_stages(G):
    from make_golden_datapath import make_info, make_scene
    from det3d.datasets.pipelines import LoadPointCloudAnnotations, LoadPointCloudFromFile
    p, _, _ = make_scene(40)
    info = make_info(7)
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "training/velodyne"))
        p[:100].tofile(os.path.join(tmp, "training/velodyne/000007.bin"))
        res = dict(metadata=dict(image_prefix=tmp, num_point_features=4), lidar={}, cam={})
        res, _ = LoadPointCloudFromFile()(res, info)
        assert np.array_equal(res["lidar"]["points"], p[:100])          # full cloud when there is no reduced one
        os.makedirs(os.path.join(tmp, "training/velodyne_reduced"))
        p.tofile(os.path.join(tmp, "training/velodyne_reduced/000007.bin"))
        res, _ = LoadPointCloudFromFile()(res, info)
        assert res["type"] == "KittiDataset" and np.array_equal(res["lidar"]["points"], p)
    res, _ = LoadPointCloudAnnotations(with_bbox=True)(res, info)
    a = res["lidar"]["annotations"]
    assert list(a["names"]) == list(G["G_names"]) and "DontCare" not in a["names"] and a["boxes"].dtype == G["G_boxes"].dtype
    assert np.allclose(a["boxes"], G["G_boxes"], rtol=0, atol=1e-5) and np.allclose(res["calib"]["frustum"], G["G_frustum"], rtol=1e-9, atol=1e-9)
    assert np.array_equal(res["cam"]["annotations"]["boxes"], G["G_cam_boxes"]) and set(res["calib"]) == {"rect", "Trv2c", "P2", "frustum"}
    d = LoadPointCloudAnnotations(with_bbox=True, enable_difficulty_level=True)(dict(type="KittiDataset", lidar={}, cam={}), info)[0]
    assert list(d["lidar"]["annotations"]["difficulty"]) == [0, 2, 0, 1, 2]


def test_voxelization_drops_ground_truth_outside_the_range():
    """the training-only filter at the head of Voxelization.__call__ (reference preprocess.py:200-206), without the device call"""
    from det3d.core.sampler import preprocess as prep
    from det3d.datasets.pipelines.preprocess import _dict_select
    from make_golden_datapath import make_scene
    _, b, n = make_scene(2)
    b[0, :2] = [90.0, 0.0]
    gt = dict(gt_boxes=b.copy(), gt_names=n.copy(), gt_classes=np.ones(len(n), np.int32))
    pcr = np.asarray([0, -40.0, -3.0, 70.4, 40.0, 1.0], np.float32)
    _dict_select(gt, prep.filter_gt_box_outside_range(gt["gt_boxes"], pcr[[0, 1, 3, 4]]))
    assert len(gt["gt_boxes"]) == len(b) - 1 == len(gt["gt_names"]) == len(gt["gt_classes"]) and np.array_equal(gt["gt_boxes"], b[1:])


def test_dataset_entry_runs_the_configured_pipeline_from_files():
    """KittiDataset.get_sensor_data -> Compose of the config's stage dicts (config.py:183-189 up to the device stages): info pickle,
    point-cloud file, database pickle + object files on disk, exactly the reference's wire format."""
    import pickle
    from make_golden_datapath import SAMPLER_CFG, make_database, make_info, make_scene, train_cfg
    from det3d.datasets.kitti.kitti import KittiDataset
    with tempfile.TemporaryDirectory() as tmp:
        infos = []
        os.makedirs(os.path.join(tmp, "training/velodyne_reduced"))
        for idx in (3, 8):
            info = make_info(idx)
            make_scene(idx)[0].tofile(os.path.join(tmp, "training/velodyne_reduced/%06d.bin" % idx))
            infos.append(info)
        with open(os.path.join(tmp, "kitti_infos_train.pkl"), "wb") as f:
            pickle.dump(infos, f)
        with open(os.path.join(tmp, "dbinfos_train.pkl"), "wb") as f:
            pickle.dump(make_database(tmp), f)
        cfg = train_cfg()
        cfg["db_sampler"] = dict(SAMPLER_CFG, db_info_path=os.path.join(tmp, "dbinfos_train.pkl"))
        pipeline = [dict(type="LoadPointCloudFromFile"), dict(type="LoadPointCloudAnnotations", with_bbox=True),
                    dict(type="Preprocess", cfg=cfg)]
        np.random.seed(9)
        ds = KittiDataset(tmp, os.path.join(tmp, "kitti_infos_train.pkl"), pipeline=pipeline, class_names=["Car"])
        assert len(ds) == 2 and ds.num_point_features == 4
        res = ds[1]
        L = res["lidar"]
        assert res["metadata"]["token"] == "8" and res["mode"] == "train" and res["labeled"] and res["type"] == "KittiDataset"
        assert set(L["annotations"]) == {"gt_boxes", "gt_names", "gt_classes"} and set(L["transformation"]) == {"flipped", "noise_rotation", "noise_scale"}
        n = len(L["annotations"]["gt_names"])
        assert n > 4 and set(L["annotations"]["gt_names"]) <= {"Car", "Van"} and L["annotations"]["gt_boxes"].shape == (n, 7)
        assert L["points"].dtype == np.float32 and L["points"].shape[1] == 4 and L["points_raw"].shape[1] == 4
        assert res["calib"]["frustum"].shape == (1, 6, 4, 3) and res["cam"]["annotations"]["boxes"].shape[1] == 4
        same = ds.get_sensor_data(8, by_index=True)
        assert same["metadata"]["image_idx"] == 8
        val = KittiDataset(tmp, os.path.join(tmp, "kitti_infos_train.pkl"), test_mode=True, class_names=["Car"], pipeline=[
            dict(type="LoadPointCloudFromFile"), dict(type="LoadPointCloudAnnotations", with_bbox=True),
            dict(type="Preprocess", cfg=dict(mode="val", shuffle_points=False, remove_environment=False, remove_unknown_examples=False))])
        r = val[0]
        assert r["mode"] == "val" and np.array_equal(r["lidar"]["points"], make_scene(3)[0]) and "transformation" not in r["lidar"]
        os.makedirs(os.path.join(tmp, "training/planes"))
        with open(os.path.join(tmp, "training/planes/000000.txt"), "w") as f:
            f.write("# Plane\nWidth 4\nHeight 1\n0.01 0.99 -0.02 -1.65\n")
        pl = val.get_road_plane(0)
        assert pl[1] < 0 and abs(np.linalg.norm(pl[:3]) - 1) < 1e-12


def test_ground_truth_database_creation_round_trip():
    """create_groundtruth_database (reference create_gt_database.py:20-131) writes what DataBaseSamplerV2 reads: crops are the
    points inside each labelled box, centre-relative; the records carry the box and the point count; the sampler pastes them back."""
    import pickle
    from make_golden_datapath import CALIB, SAMPLER_CFG, make_scene
    from det3d.builder import build_dbsampler
    from det3d.core.bbox import box_np_ops
    from det3d.datasets.utils.create_gt_database import create_groundtruth_database
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "training/velodyne_reduced"))
        infos, scenes = [], {}
        for idx in (1, 2, 5):
            pts, boxes, names = make_scene(idx)
            pts.tofile(os.path.join(tmp, "training/velodyne_reduced/%06d.bin" % idx))
            # lidar boxes -> the camera-frame annotation the info file holds (bottom-centre location, l,h,w sizes)
            low = boxes.copy(); low[:, 2] -= low[:, 5] / 2
            cam = box_np_ops.box_lidar_to_camera(low.astype(np.float64), CALIB["R0_rect"], CALIB["Tr_velo_to_cam"])
            annos = dict(name=names.copy(), location=cam[:, :3], dimensions=cam[:, 3:6], rotation_y=cam[:, 6],
                         bbox=np.zeros((len(names), 4)), difficulty=np.arange(len(names)) % 3)
            infos.append(dict(image=dict(image_idx=idx, image_shape=np.array([375, 1242], np.int32)), calib=dict(CALIB), annos=annos,
                              point_cloud=dict(num_features=4, velodyne_path="training/velodyne/%06d.bin" % idx)))
            scenes[idx] = (pts, boxes, names)
        with open(os.path.join(tmp, "kitti_infos_train.pkl"), "wb") as f:
            pickle.dump(infos, f)
        db = create_groundtruth_database("KITTI", tmp, os.path.join(tmp, "kitti_infos_train.pkl"))
        assert set(db) == {"Car", "Pedestrian", "Van"} and len(db["Car"]) == 15 and len(db["Van"]) == 3
        assert pickle.load(open(os.path.join(tmp, "dbinfos_train.pkl"), "rb")).keys() == db.keys()
        assert sorted(r["group_id"] for v in db.values() for r in v) == list(range(21))
        rec = db["Car"][7]
        pts, boxes, names = scenes[rec["image_idx"]]
        assert np.allclose(rec["box3d_lidar"], boxes[rec["gt_idx"]], atol=1e-4) and rec["path"].startswith("gt_database/")
        obj = np.fromfile(os.path.join(tmp, rec["path"]), dtype=np.float32).reshape(-1, 4)
        assert obj.shape[0] == rec["num_points_in_gt"] and abs(obj.shape[0] - 330) <= 3   # the cluster the scene put into the box
        back = obj.copy(); back[:, :3] += rec["box3d_lidar"][:3]
        assert box_np_ops.points_in_rbbox(back, rec["box3d_lidar"][None]).all()
        # ... and the sampler consumes it
        np.random.seed(3)
        sampler = build_dbsampler(dict(SAMPLER_CFG, db_prep_steps=[dict(filter_by_min_num_points=dict(Car=5))]), db_infos=db)
        _, b, n = make_scene(77)
        got = sampler.sample_all(tmp, b, n, 4)
        assert got is not None and set(got["gt_names"]) <= {"Car", "Van"} and got["points"].shape[1] == 4
        assert box_np_ops.points_in_rbbox(got["points"], got["gt_boxes"]).any(1).all()
        # widened crops go to their own files
        wide = create_groundtruth_database("KITTI", tmp, os.path.join(tmp, "kitti_infos_train.pkl"), used_classes=["Car"], gt_aug_with_context=0.5)
        assert set(wide) == {"Car"} and os.path.exists(os.path.join(tmp, "dbinfos_enlarged_train.pkl"))
        w0 = np.fromfile(os.path.join(tmp, wide["Car"][7]["path"]), dtype=np.float32).reshape(-1, 4)
        assert w0.shape[0] >= obj.shape[0] and wide["Car"][7]["num_points_in_gt"] == rec["num_points_in_gt"]


def test_preprocess_with_empty_or_untargeted_ground_truth():
    """frames without labelled cars: no boxes at all, or only classes outside the target list; with and without the sampler"""
    from make_golden_datapath import SAMPLER_CFG, make_database, make_scene, train_cfg
    from det3d.builder import build_dbsampler
    from det3d.datasets.pipelines import Preprocess
    from det3d.datasets.utils import sa_da_v2
    pts, b, n = make_scene(3)
    assert sa_da_v2.pyramid_augment_v0(b[:0], pts.copy()).shape == pts.shape
    frame = lambda tmp, bb, nn: dict(labeled=True, metadata=dict(image_prefix=tmp, num_point_features=4),
                                     lidar=dict(points=pts.copy(), annotations=dict(boxes=bb.copy(), names=nn.copy())))
    with tempfile.TemporaryDirectory() as tmp:
        np.random.seed(1)
        stage = Preprocess(cfg=train_cfg(), db_sampler=build_dbsampler(dict(SAMPLER_CFG), db_infos=make_database(tmp)))
        res, _ = stage(frame(tmp, b[:0], n[:0]), None)
        a = res["lidar"]["annotations"]
        assert len(a["gt_names"]) > 5 and a["gt_boxes"].shape == (len(a["gt_names"]), 7) and res["lidar"]["points"].shape[0] > pts.shape[0]
        res, _ = stage(frame(tmp, b[5:6], n[5:6]), None)     # a lone pedestrian blocks pasted cars but is not a target itself
        assert "Pedestrian" not in res["lidar"]["annotations"]["gt_names"] and len(res["lidar"]["annotations"]["gt_names"]) > 5
        bare = Preprocess(cfg=train_cfg())
        res, _ = bare(frame(tmp, b[:0], n[:0]), None)
        assert res["lidar"]["annotations"]["gt_boxes"].shape == (0, 7) and res["lidar"]["points"].shape == pts.shape
        assert res["lidar"]["annotations"]["gt_classes"].dtype == np.int32 and res["lidar"]["annotations_raw"]["gt_boxes"].shape == (0, 7)
