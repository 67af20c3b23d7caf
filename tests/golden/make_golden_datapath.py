"""Generates tests/golden/datapath_ref.npz by running the REFERENCE's training data path from source on CPU:
    det3d/core/bbox/geometry.py            points_in_convex_polygon_3d_jit :215-276, points_in_convex_polygon_jit :279-325
    det3d/core/bbox/box_np_ops.py          points_in_rbbox :1152, box2d_to_corner_jit :535, box_camera_to_lidar :965, ...
    det3d/core/sampler/preprocess.py       box_collision_test :944, noise_per_box :579, noise_per_object_v4_ :615,
                                           random_flip_v2 :896, global_rotation_v3 :930, global_scaling_v3 :914, BatchSampler :20,
                                           filter_gt_box_outside_range :138
    det3d/core/sampler/sample_ops_v2.py    DataBaseSamplerV2.sample_all :62-196 (GT-AUG)
    det3d/datasets/utils/sa_da_v2.py       pyramid_augment_v0 :76-205 (shape-aware augmentation)
    det3d/datasets/pipelines/preprocess.py Preprocess.__call__ :62-175
    det3d/datasets/pipelines/loading.py    LoadPointCloudFromFile / LoadPointCloudAnnotations :73-160
numba is absent: the kernels run as Python loops. Three places where CPython and compiled numba differ are bridged so that the
golden vectors describe the COMPILED behaviour the reference ships:
  * box_collision_test tests `ret[i, j] is True / is False` on a numpy bool; numba compiles `is` between booleans as a value
    comparison, CPython's identity test is never true and would skip the containment branch -> the source is executed with
    those three tests rewritten to `==`;
  * surface_equ_3d_jitv2 reads surfaces[0, 0, 0] before its loops; with zero polygons numba reads (unchecked) garbage it never
    uses, CPython raises -> an empty-input guard is wrapped around it;
  * sa_da_v2 imports `ifp` (jackd/ifp-sample, external, absent): stubbed with the restatement in the mirror, so the thinning
    step's point choice is unpinned (everything around it is pinned).
Run in the build container only:  python tests/golden/make_golden_datapath.py"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CALIB = dict(
    R0_rect=np.array([[0.9999239, 0.00983776, -0.00744505, 0.], [-0.0098698, 0.9999421, -0.00427846, 0.],
                      [0.00740253, 0.00435161, 0.9999631, 0.], [0., 0., 0., 1.]]),
    Tr_velo_to_cam=np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
                             [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                             [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01], [0., 0., 0., 1.]]),
    P2=np.array([[721.5377, 0., 609.5593, 44.85728], [0., 721.5377, 172.854, 0.2163791], [0., 0., 1., 0.002745884], [0., 0., 0., 1.]]))


def car_boxes(rng, n, x0=6.0, x1=62.0):
    """n car-sized boxes on a jittered grid (no two overlap), [x, y, z, w, l, h, r] float32."""
    cells = [(gx, gy) for gx in range(7) for gy in range(6)]
    pick = rng.choice(len(cells), n, replace=False)
    b = np.zeros((n, 7), np.float32)
    for k, c in enumerate(pick):
        gx, gy = cells[c]
        b[k, 0] = x0 + gx * (x1 - x0) / 7 + rng.uniform(0.5, 2.0)
        b[k, 1] = -33.0 + gy * 11.0 + rng.uniform(0.5, 3.0)
    b[:, 2] = rng.uniform(-1.1, -0.7, n)
    b[:, 3] = rng.uniform(1.5, 1.8, n); b[:, 4] = rng.uniform(3.4, 4.4, n); b[:, 5] = rng.uniform(1.4, 1.7, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def fill_box(rng, box, n):
    """n points uniformly inside the box (slightly shrunk), with intensity: (n, 4) float32 in the lidar frame."""
    u = rng.uniform(-0.48, 0.48, (n, 3)) * box[3:6]
    s, c = np.sin(box[6]), np.cos(box[6])
    p = np.stack([u[:, 0] * c + u[:, 1] * s, -u[:, 0] * s + u[:, 1] * c, u[:, 2]], axis=1) + box[:3]
    return np.concatenate([p, rng.uniform(0, 1, (n, 1))], axis=1).astype(np.float32)


def make_scene(seed, n_gt=7, n_bg=1200, per_box=330):
    """a frame: background points on a rough ground plane + dense clusters inside the labelled boxes (5 cars, a pedestrian, a van)."""
    rng = np.random.RandomState(seed)
    boxes = car_boxes(rng, n_gt)
    names = np.array(["Car"] * (n_gt - 2) + ["Pedestrian", "Van"])
    boxes[n_gt - 2, 3:6] = [0.6, 0.8, 1.75]
    bg = np.stack([rng.uniform(0, 70, n_bg), rng.uniform(-40, 40, n_bg), rng.normal(-1.7, 0.05, n_bg), rng.uniform(0, 1, n_bg)], 1)
    pts = np.concatenate([bg.astype(np.float32)] + [fill_box(rng, b, per_box) for b in boxes], axis=0)
    return pts[rng.permutation(pts.shape[0])], boxes, names


def make_database(root, seed=11, n_car=40, n_van=6):
    """a ground-truth database: per class a list of dict(name, path, box3d_lidar, num_points_in_gt, difficulty, ...) and one point
    file per object holding centre-relative float32 x,y,z,intensity (what create_groundtruth_database writes)."""
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "gt_database"), exist_ok=True)
    db = {"Car": [], "Van": [], "Pedestrian": []}
    for cls, n in (("Car", n_car), ("Van", n_van), ("Pedestrian", 3)):
        for k, box in enumerate(car_boxes(rng, n)):
            npts = int(rng.choice([3, 40, 120, 260]))
            pts = fill_box(rng, box, npts)
            pts[:, :3] -= box[:3]
            rel = "gt_database/%s_%d.bin" % (cls, k)
            pts.tofile(os.path.join(root, rel))
            db[cls].append(dict(name=cls, path=rel, image_idx=k, gt_idx=k, box3d_lidar=box, num_points_in_gt=npts,
                                difficulty=int(rng.choice([-1, 0, 1, 2])), group_id=k))
    return db


SAMPLER_CFG = dict(type="GT-AUG", enable=True, db_info_path="unused", sample_groups=[dict(Car=15)],
                   db_prep_steps=[dict(filter_by_min_num_points=dict(Car=5)), dict(filter_by_difficulty=[-1])],
                   global_random_rotation_range_per_object=[0, 0], rate=1.0, gt_random_drop=-1.0, gt_aug_with_context=-1.0,
                   gt_aug_similar_type=True)


def train_cfg():
    """config.py:140-165 with the my_paras defaults of the SE-SSD run."""
    return dict(mode="train", shuffle_points=True, gt_loc_noise=[1.0, 1.0, 0.5], gt_rot_noise=[-0.785, 0.785],
                global_rot_noise=[-0.785, 0.785], global_scale_noise=[0.95, 1.05], global_rot_per_obj_range=[0, 0],
                global_trans_noise=[0.0, 0.0, 0.0], remove_points_after_sample=True, gt_drop_percentage=0.0,
                gt_drop_max_keep_points=15, remove_environment=False, remove_unknown_examples=False, class_names=["Car"],
                symmetry_intensity=False, enable_similar_type=True, min_points_in_gt=-1, data_aug_with_context=-1.0,
                data_aug_random_drop=-1.0)


def make_info(seed):
    """a kitti_infos entry: camera-frame annotations (with a DontCare row), calibration, image shape."""
    rng = np.random.RandomState(seed)
    n = 6
    loc = np.stack([rng.uniform(-15, 15, n), rng.uniform(1.2, 1.9, n), rng.uniform(6, 55, n)], 1)
    dims = np.stack([rng.uniform(3.4, 4.4, n), rng.uniform(1.4, 1.7, n), rng.uniform(1.5, 1.8, n)], 1)   # l, h, w
    annos = dict(name=np.array(["Car", "DontCare", "Car", "Pedestrian", "Van", "Car"]), location=loc, dimensions=dims,
                 rotation_y=rng.uniform(-np.pi, np.pi, n), bbox=rng.uniform(0, 300, (n, 4)), difficulty=np.arange(n) % 3,
                 truncated=np.zeros(n), occluded=np.zeros(n, np.int64), alpha=rng.uniform(-3, 3, n))
    return dict(image=dict(image_idx=seed, image_shape=np.array([375, 1242], np.int32)), calib=dict(CALIB), annos=annos,
                point_cloud=dict(num_features=4, velodyne_path="training/velodyne/%06d.bin" % seed))


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def load_reference():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
    sys.path.insert(0, HERE)
    import make_golden as MG
    assert os.path.isdir(MG.REF)
    # the mirror's farthest-point restatement, taken BEFORE the det3d.* names are rebound to the reference's files
    import importlib.util
    spec = importlib.util.spec_from_file_location("mirror_sa_da_v2", os.path.join(ROOT, "se-ssd_amd/det3d/datasets/utils/sa_da_v2.py"))
    MG.install_stubs()
    for name in ("det3d.core.sampler", "det3d.core.evaluation", "det3d.core.evaluation.bbox_overlaps", "det3d.core.input",
                 "det3d.core.input.voxel_generator", "det3d.core.anchor", "det3d.core.anchor.target_assigner", "det3d.builder",
                 "det3d.torchie", "det3d.datasets", "det3d.datasets.kitti", "det3d.datasets.utils", "det3d.datasets.pipelines",
                 "det3d.datasets.registry", "det3d.utils.check", "ifp", "ipdb", "skimage", "tqdm", "pycocotools", "pycocotools.mask"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["det3d"].torchie = sys.modules["det3d.torchie"]
    sys.modules["skimage"].io = None
    sys.modules["tqdm"].tqdm = lambda x, **k: x
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]

    def _no_debugger():
        raise RuntimeError("reference dropped into its debugger")
    sys.modules["ipdb"].set_trace = _no_debugger
    geo = MG.load_ref("det3d/core/bbox/geometry.py", "det3d.core.bbox.geometry")
    raw_equ = geo.surface_equ_3d_jitv2

    def guarded_equ(surfaces):
        if surfaces.shape[0] == 0:
            return np.zeros((0, surfaces.shape[1], 3), surfaces.dtype), np.zeros((0, surfaces.shape[1]), surfaces.dtype)
        return raw_equ(surfaces)
    geo.surface_equ_3d_jitv2 = guarded_equ
    bnp = MG.load_ref("det3d/core/bbox/box_np_ops.py", "det3d.core.bbox.box_np_ops")
    sys.modules["det3d.core.bbox"].box_np_ops = bnp
    sys.modules["det3d.core.bbox"].geometry = geo
    # sampler/preprocess.py with numba's meaning of `is True` / `is False` on booleans
    src = open(os.path.join(MG.REF, "det3d/core/sampler/preprocess.py")).read()
    assert src.count("is True") == 1 and src.count("is False") == 4
    src = src.replace(" is True", " == True").replace(" is False", " == False")
    prep = types.ModuleType("det3d.core.sampler.preprocess")
    prep.__file__ = os.path.join(MG.REF, "det3d/core/sampler/preprocess.py")
    sys.modules["det3d.core.sampler.preprocess"] = prep
    exec(compile(src.split('if __name__ == "__main__":')[0], prep.__file__, "exec"), prep.__dict__)
    sys.modules["det3d.core.sampler"].preprocess = prep
    sys.modules["det3d.utils.check"].shape_mergeable = lambda x, s: True
    sops = MG.load_ref("det3d/core/sampler/sample_ops_v2.py", "det3d.core.sampler.sample_ops_v2")
    mirror_sa = importlib.util.module_from_spec(spec)
    # executing the mirror module needs the mirror's det3d.core.bbox names: give it the reference's (same API)
    spec.loader.exec_module(mirror_sa)
    sys.modules["ifp"].ifp_sample = mirror_sa.ifp_sample
    sada = MG.load_ref("det3d/datasets/utils/sa_da_v2.py", "det3d.datasets.utils.sa_da_v2")
    sys.modules["det3d.datasets.utils"].sa_da_v2 = sada
    kc = MG.load_ref("det3d/datasets/kitti/kitti_common.py", "det3d.datasets.kitti.kitti_common")
    sys.modules["det3d.datasets.kitti"].kitti_common = kc

    class _Reg:
        @staticmethod
        def register_module(cls):
            return cls
    sys.modules["det3d.datasets.registry"].PIPELINES = _Reg
    sys.modules["det3d.core.evaluation.bbox_overlaps"].bbox_overlaps = None
    sys.modules["det3d.core.input.voxel_generator"].VoxelGenerator = None
    sys.modules["det3d.core.anchor.target_assigner"].TargetAssigner = None
    import logging
    state = {}

    def build_dbsampler(cfg, logger=None):   # builder.py:378-406 without the pickle read
        prepors = []
        for c in cfg["db_prep_steps"]:
            if "filter_by_difficulty" in c:
                prepors.append(prep.DBFilterByDifficulty(c["filter_by_difficulty"], logger=logging.getLogger("x")))
            else:
                prepors.append(prep.DBFilterByMinNumPoint(c["filter_by_min_num_points"], logger=logging.getLogger("x")))
        return sops.DataBaseSamplerV2(state["db"], cfg["sample_groups"], prep.DataBasePreprocessor(prepors), cfg["rate"],
                                      list(cfg["global_random_rotation_range_per_object"]), logger=logging.getLogger("x"),
                                      gt_random_drop=cfg["gt_random_drop"], gt_aug_with_context=cfg["gt_aug_with_context"],
                                      gt_aug_similar_type=cfg["gt_aug_similar_type"])
    b = sys.modules["det3d.builder"]
    b.build_dbsampler, b.build_anchor_generator, b.build_similarity_metric, b.build_box_coder = build_dbsampler, None, None, None
    pipe = MG.load_ref("det3d/datasets/pipelines/preprocess.py", "det3d.datasets.pipelines.preprocess")
    load = MG.load_ref("det3d/datasets/pipelines/loading.py", "det3d.datasets.pipelines.loading")
    return dict(geo=geo, bnp=bnp, prep=prep, sops=sops, sada=sada, pipe=pipe, load=load, state=state, build_dbsampler=build_dbsampler)


def containment_case():
    """BEV quadrilaterals with an edge crossing, a strict containment, a bounding-rectangle-only overlap and a far pair."""
    def quad(cx, cy, w, l, r):
        u = np.array([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]]) * [w, l]
        s, c = np.sin(r), np.cos(r)
        return np.stack([u[:, 0] * c + u[:, 1] * s, -u[:, 0] * s + u[:, 1] * c], 1) + [cx, cy]
    return np.stack([quad(0, 0, 4, 8, 0.3), quad(0.2, 0.1, 1, 2, 1.0), quad(2.5, 3.0, 2, 4, -0.6), quad(3.6, -4.2, 1.5, 1.5, 0.78),
                     quad(30, 30, 2, 4, 0.0)])


def main():
    R = load_reference()
    bnp, prep, sops, sada, pipe, load = R["bnp"], R["prep"], R["sops"], R["sada"], R["pipe"], R["load"]
    out = {}
    # --- A. primitives
    pts, boxes, names = make_scene(1)
    out["A_in_rbbox"] = np.packbits(bnp.points_in_rbbox(pts, boxes))
    quads = containment_case()
    out["A_collision"] = prep.box_collision_test(quads, quads)
    corners = bnp.center_to_corner_box2d(boxes[:, :2], boxes[:, 3:5], boxes[:, 6])
    out["A_collision_scene"] = prep.box_collision_test(corners, corners)
    edge = boxes.copy(); edge[0, :2] = [70.2, 39.9]; edge[1, :2] = [72.9, 0.0]; edge[2, :2] = [-2.6, 10.0]
    out["A_range_mask"] = prep.filter_gt_box_outside_range(edge, np.array([0, -40.0, 70.4, 40.0], np.float32))
    out["A_center_mask"] = prep.filter_gt_box_outside_range_by_center(edge, np.array([0, -40.0, 70.4, 40.0], np.float32))
    rng = np.random.RandomState(3)
    crowded = boxes[:, [0, 1, 3, 4, 6]].astype(np.float64)
    crowded[1, :2] = crowded[0, :2] + [2.2, 0.5]       # neighbours close enough that many candidate moves collide
    crowded[2, :2] = crowded[0, :2] + [-2.0, 1.0]
    loc, rot = rng.normal(scale=[1.0, 1.0, 0.5], size=(len(boxes), 30, 3)), rng.uniform(-0.785, 0.785, (len(boxes), 30))
    valid = np.array([n in ("Car", "Van") for n in names])
    out["A_noise_boxes"], out["A_noise_loc"], out["A_noise_rot"], out["A_noise_valid"] = crowded, loc, rot, valid
    out["A_noise_chosen"] = prep.noise_per_box(crowded.copy(), valid, loc, rot)
    # --- B. per-object noise
    for seed in (0, 1):
        p, b = pts.copy(), boxes.copy()
        np.random.seed(100 + seed)
        prep.noise_per_object_v4_(b, p, valid, rotation_perturb=[-0.785, 0.785], center_noise_std=[1.0, 1.0, 0.5],
                                  global_random_rot_range=[0, 0], group_ids=None, num_try=100, data_aug_with_context=-1.0,
                                  data_aug_random_drop=-1.0)
        out["B%d_boxes" % seed], out["B%d_points" % seed] = b, p
    # --- C. global transforms
    for seed in range(4):
        p, b = pts.copy(), boxes.copy()
        np.random.seed(200 + seed)
        b, p, f = prep.random_flip_v2(b, p)
        b, p, r = prep.global_rotation_v3(b, p, [-0.785, 0.785])
        b, p, s = prep.global_scaling_v3(b, p, 0.95, 1.05)
        out["C%d_boxes" % seed], out["C%d_points" % seed], out["C%d_t" % seed] = b, p, np.array([float(f), r, s])
    # --- D. shape-aware augmentation: stage by stage, then the configured mix
    cars = boxes[valid]
    for tag, kw, seed in (("drop", dict(enable_sa_dropout=0.6, enable_sa_sparsity=None, enable_sa_swap=None), 300),
                          ("sparse", dict(enable_sa_dropout=None, enable_sa_sparsity=[0.7, 30], enable_sa_swap=None), 301),
                          ("swap", dict(enable_sa_dropout=None, enable_sa_sparsity=None, enable_sa_swap=[0.7, 20]), 302),
                          ("mix", dict(enable_sa_dropout=0.25, enable_sa_sparsity=[0.05, 50], enable_sa_swap=[0.1, 50]), 303),
                          ("mix2", dict(enable_sa_dropout=0.4, enable_sa_sparsity=[0.4, 30], enable_sa_swap=[0.5, 20]), 304)):
        np.random.seed(seed)
        out["D_" + tag] = sada.pyramid_augment_v0(cars.copy(), pts.copy(), **kw)
        print("sa-da", tag, pts.shape[0], "->", out["D_" + tag].shape[0])
    out["D_pyramids"] = sada.get_pyramids(cars)
    # --- E. GT-AUG sampler
    with tempfile.TemporaryDirectory() as tmp:
        R["state"]["db"] = make_database(tmp)
        np.random.seed(400)
        sampler = R["build_dbsampler"](SAMPLER_CFG)
        for k in range(3):
            _, b, n = make_scene(20 + k)
            got = sampler.sample_all(tmp, b, n, 4, False, gt_group_ids=None, calib=None, targeted_class_names=["Car", "Van"])
            out["E%d_boxes" % k], out["E%d_points" % k], out["E%d_names" % k] = got["gt_boxes"], got["points"], got["gt_names"]
            print("gt-aug", k, "pasted", len(got["gt_names"]))
        # --- F. the whole Preprocess stage
        R["state"]["db"] = make_database(tmp)
        cfg = AttrDict(train_cfg()); cfg["db_sampler"] = AttrDict(SAMPLER_CFG)
        np.random.seed(500)
        stage = pipe.Preprocess(cfg=cfg)
        for k in range(2):
            p, b, n = make_scene(30 + k)
            res = dict(labeled=True, metadata=dict(image_prefix=tmp, num_point_features=4),
                       lidar=dict(points=p, annotations=dict(boxes=b, names=n)))
            res, _ = stage(res, None)
            L = res["lidar"]
            out["F%d_points" % k], out["F%d_points_raw" % k] = L["points"], L["points_raw"]
            out["F%d_boxes" % k], out["F%d_boxes_raw" % k] = L["annotations"]["gt_boxes"], L["annotations_raw"]["gt_boxes"]
            out["F%d_names" % k], out["F%d_classes" % k] = L["annotations"]["gt_names"], L["annotations"]["gt_classes"]
            t = L["transformation"]
            out["F%d_t" % k] = np.array([float(t["flipped"]), t["noise_rotation"], t["noise_scale"]])
            print("preprocess", k, p.shape[0], "->", L["points"].shape[0], "boxes", len(b), "->", len(L["annotations"]["gt_boxes"]))
        p, b, n = make_scene(40)
        res, _ = stage(dict(labeled=False, metadata=dict(image_prefix=tmp, num_point_features=4), lidar=dict(points=p)), None)
        t = res["lidar"]["transformation"]
        out["F_unlabeled_points"], out["F_unlabeled_t"] = res["lidar"]["points"], np.array([float(t["flipped"]), t["noise_rotation"], t["noise_scale"]])
        val = pipe.Preprocess(cfg=AttrDict(mode="val", shuffle_points=False, remove_environment=False, remove_unknown_examples=False))
        res, _ = val(dict(labeled=False, lidar=dict(points=p.copy())), None)
        assert res["mode"] == "val" and np.array_equal(res["lidar"]["points"], p)
        # --- G. loading
        info = make_info(7)
        os.makedirs(os.path.join(tmp, "training/velodyne_reduced"))
        os.makedirs(os.path.join(tmp, "training/velodyne"))
        p[:100].tofile(os.path.join(tmp, "training/velodyne/000007.bin"))
        p.tofile(os.path.join(tmp, "training/velodyne_reduced/000007.bin"))
        res = dict(metadata=dict(image_prefix=tmp, num_point_features=4), lidar={}, cam={})
        res, _ = load.LoadPointCloudFromFile()(res, info)
        assert np.array_equal(res["lidar"]["points"], p)
        res, _ = load.LoadPointCloudAnnotations(with_bbox=True)(res, info)
        out["G_boxes"], out["G_names"] = res["lidar"]["annotations"]["boxes"], res["lidar"]["annotations"]["names"]
        out["G_frustum"], out["G_cam_boxes"] = res["calib"]["frustum"], res["cam"]["annotations"]["boxes"]
    np.savez_compressed(os.path.join(HERE, "datapath_ref.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
