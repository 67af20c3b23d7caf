"""mirrors det3d/datasets/pipelines/preprocess.py: Preprocess (:30-175, the augmentation stage), Voxelization (:178-232, the
stage that feeds the hot path) and AssignTarget (:236-358).
Voxelization: same result dict (`res["lidar"]["voxels"]` = voxels / coordinates / num_points / num_voxels / shape, plus the
`*_raw` twin when `points_raw` is present), voxelized on the MI355X through VoxelGenerator.generate -> sessd_voxelize_frame."""
import numpy as np

from det3d.core.input.voxel_generator import VoxelGenerator

from ..registry import PIPELINES


def _get(cfg, key, default=None):
    return cfg[key] if key in cfg else default


def _dict_select(dict_, inds):
    """in place: every array of the (nested) dict indexed by inds / a mask (:19-27)."""
    for k, v in dict_.items():
        if isinstance(v, dict):
            _dict_select(v, inds)
        else:
            dict_[k] = v[inds]


@PIPELINES.register_module
class Preprocess(object):
    """The augmentation stage of the training pipeline and its pass-through validation mode (:30-175), host side as in the
    reference. Labeled training frames: drop DontCare, paste database objects (GT-AUG) and remove the scene points they
    cover, per-object noise, snapshot (`points_raw`, `annotations_raw`: the teacher's view), then global flip / rotation /
    scaling recorded in `transformation` (what MultiGroupHead.consistency_loss undoes), shape-aware augmentation, point shuffle.
    With a CUDA tensor as `points` the point-level work of the same stage runs on the device (`_call_device`).
    Unlabeled training frames only get the global transformation. Options the SE-SSD configuration leaves off
    (remove_environment, remove_unknown, min_points_in_gt, rgb, reference detections, random_crop, npoints) are not mirrored."""

    def __init__(self, cfg=None, db_sampler=None, **kwargs):
        self.shuffle_points = cfg["shuffle_points"]
        self.mode = cfg["mode"]
        for key in ("remove_environment", "remove_unknown_examples", "add_rgb_to_points", "reference_detections",
                    "remove_outside_points", "random_crop", "random_select", "symmetry_intensity"):
            if _get(cfg, key, False):
                raise NotImplementedError("Preprocess option %s is off in the SE-SSD configuration and not mirrored" % key)
        if self.mode == "train":
            self.gt_loc_noise_std = cfg["gt_loc_noise"]
            self.gt_rotation_noise = cfg["gt_rot_noise"]
            self.global_rotation_noise = cfg["global_rot_noise"]
            self.global_scaling_noise = cfg["global_scale_noise"]
            self.global_random_rot_range = cfg["global_rot_per_obj_range"]
            self.remove_points_after_sample = cfg["remove_points_after_sample"]
            self.class_names = cfg["class_names"]
            self.enable_similar_type = _get(cfg, "enable_similar_type", False)
            if self.enable_similar_type and "Car" in self.class_names and "Van" not in self.class_names:
                self.class_names.append("Van")   # the reference appends to the config's own list (:54-55)
            if db_sampler is not None:
                self.db_sampler = db_sampler
            elif _get(cfg, "db_sampler", None):
                from det3d.builder import build_dbsampler
                self.db_sampler = build_dbsampler(cfg["db_sampler"])
            else:
                self.db_sampler = None
            self.data_aug_with_context = _get(cfg, "data_aug_with_context", -1)
            self.data_aug_random_drop = _get(cfg, "data_aug_random_drop", -1)
            self.sa_da = dict(enable_sa_dropout=0.25, enable_sa_sparsity=[0.05, 50], enable_sa_swap=[0.1, 50])  # cars (:134-138)

    def _global_device(self, gt_boxes, points, snapshot):
        """The three global draws in the host stage's order, applied to the boxes on the host and to the DEVICE cloud in one launch
        (sessd_points_global_transform), which also leaves the untransformed cloud in `snapshot` (points_raw)."""
        from det3d.core.sampler import preprocess as prep
        from sessd_hip import ops
        none = np.zeros((0, 4), np.float32)
        gt_boxes, _, flipped = prep.random_flip_v2(gt_boxes, none)
        gt_boxes, _, rot = prep.global_rotation_v3(gt_boxes, none, self.global_rotation_noise)
        gt_boxes, _, scale = prep.global_scaling_v3(gt_boxes, none, *self.global_scaling_noise)
        ops.points_global_transform_(points, bool(flipped), float(rot), float(scale), raw_copy=snapshot)
        return gt_boxes, {"flipped": flipped, "noise_rotation": rot, "noise_scale": scale}

    def _call_device(self, res, info):
        """The stage with the point cloud resident on the device (res["lidar"]["points"] is a CUDA tensor): box-level decisions and
        every random draw are the host stage's, in its order (a seed gives the same augmentation); the point-level work runs on
        sessd_points_in_bodies / _compact (GT-AUG removal), _rigid_moves (per-object noise), _global_transform (+ the points_raw
        snapshot), the shape-aware augmentation (sa_da_v2.pyramid_augment_v0_device: membership, removal, farthest-point thinning
        and the pyramid swap on the device) and a device gather (shuffle). What crosses to the host are per-pyramid point counts and
        row counts of compactions."""
        import torch
        from det3d.core.bbox import box_np_ops
        from det3d.core.bbox.geometry import surface_equ_3d_jitv2
        from det3d.core.sampler import preprocess as prep
        from det3d.datasets.kitti import kitti_common as kitti
        from det3d.datasets.utils import sa_da_v2
        from sessd_hip import ops
        res["mode"] = self.mode
        points = res["lidar"]["points"].float().contiguous()
        dev = points.device

        def planes_of(surfaces):
            nrm, d = surface_equ_3d_jitv2(surfaces[:, :, :3, :])
            return torch.from_numpy(np.ascontiguousarray(np.concatenate([nrm, d[..., None]], axis=-1).astype(np.float32))).to(dev)

        labeled = self.mode == "train" and res["labeled"]
        if labeled:
            anno = res["lidar"]["annotations"]
            gt_dict = {"gt_boxes": anno["boxes"], "gt_names": np.array(anno["names"]).reshape(-1)}
            _dict_select(gt_dict, kitti.drop_arrays_by_name(gt_dict["gt_names"], ["DontCare", "ignore"]))
            target = np.array([n in self.class_names for n in gt_dict["gt_names"]], dtype=np.bool_)
            if self.db_sampler:
                pasted = self.db_sampler.sample_all(res["metadata"]["image_prefix"], gt_dict["gt_boxes"], gt_dict["gt_names"],
                                                    res["metadata"]["num_point_features"], False, gt_group_ids=None,
                                                    calib=res["calib"] if "calib" in res else None,
                                                    targeted_class_names=self.class_names)
                if pasted is not None:
                    gt_dict["gt_names"] = np.concatenate([gt_dict["gt_names"], pasted["gt_names"]], axis=0)
                    gt_dict["gt_boxes"] = np.concatenate([gt_dict["gt_boxes"], pasted["gt_boxes"]])
                    target = np.concatenate([target, pasted["gt_masks"]], axis=0)
                    if self.remove_points_after_sample and pasted["gt_boxes"].shape[0] and points.shape[0]:
                        pb = pasted["gt_boxes"]
                        corners = box_np_ops.center_to_corner_box3d(pb[:, :3], pb[:, 3:6], pb[:, 6], origin=(0.5, 0.5, 0.5), axis=2)
                        inside = ops.points_in_bodies(points, planes_of(box_np_ops.corner_to_surfaces_3d(corners)))
                        kept, n_kept = ops.points_compact(points, ~inside.any(-1))
                        points = kept[: int(n_kept.item())]
                    points = torch.cat([torch.from_numpy(np.ascontiguousarray(pasted["points"], np.float32)).to(dev), points], dim=0).contiguous()

            def move(surfaces, centers, loc_t, rot_t, valid):
                if points.shape[0] and 0 < surfaces.shape[0] <= 128:
                    ops.points_rigid_moves_(points, planes_of(surfaces), centers, loc_t, rot_t, valid)
                elif surfaces.shape[0] > 128:
                    # more boxes than sessd_points_rigid_moves holds in LDS (128): membership must come from the UNMOVED cloud for
                    # all boxes at once, so the call is not chunked -- this rare frame takes the host function and goes back
                    ph = points.cpu().numpy()
                    from det3d.core.bbox.geometry import points_in_convex_polygon_3d_jit
                    masks = points_in_convex_polygon_3d_jit(ph[:, :3], surfaces)
                    prep.points_transform_(ph, centers, masks, loc_t, rot_t, valid)
                    points.copy_(torch.from_numpy(ph).to(dev))

            prep.noise_per_object_v4_(gt_dict["gt_boxes"], move, target, rotation_perturb=self.gt_rotation_noise,
                                      center_noise_std=self.gt_loc_noise_std, global_random_rot_range=self.global_random_rot_range,
                                      group_ids=None, num_try=100, data_aug_with_context=self.data_aug_with_context,
                                      data_aug_random_drop=self.data_aug_random_drop)
            _dict_select(gt_dict, target)
            gt_dict["gt_classes"] = np.array([self.class_names.index(n) + 1 for n in gt_dict["gt_names"]], dtype=np.int32)
            raw = torch.empty_like(points)
            res["lidar"]["annotations_raw"] = {k: v.copy() for k, v in gt_dict.items()}
            gt_dict["gt_boxes"], res["lidar"]["transformation"] = self._global_device(gt_dict["gt_boxes"], points, raw)
            res["lidar"]["points_raw"] = raw
            points = sa_da_v2.pyramid_augment_v0_device(gt_dict["gt_boxes"], points, **self.sa_da)
        if self.shuffle_points:
            perm = np.random.choice(np.arange(points.shape[0]), points.shape[0], replace=False)
            points = points[torch.from_numpy(perm).to(dev)]
        if self.mode == "train" and not res["labeled"]:
            points = points.contiguous()
            _, res["lidar"]["transformation"] = self._global_device(None, points, None)
        res["lidar"]["points"] = points
        if labeled:
            res["lidar"]["annotations"] = gt_dict
        return res, info

    def _global(self, gt_boxes, points):
        from det3d.core.sampler import preprocess as prep
        gt_boxes, points, flipped = prep.random_flip_v2(gt_boxes, points)
        gt_boxes, points, rot = prep.global_rotation_v3(gt_boxes, points, self.global_rotation_noise)
        gt_boxes, points, scale = prep.global_scaling_v3(gt_boxes, points, *self.global_scaling_noise)
        return gt_boxes, points, {"flipped": flipped, "noise_rotation": rot, "noise_scale": scale}

    def __call__(self, res, info):
        from det3d.core.bbox import box_np_ops
        from det3d.core.sampler import preprocess as prep
        from det3d.datasets.kitti import kitti_common as kitti
        from det3d.datasets.utils import sa_da_v2
        points = res["lidar"]["points"]
        if not isinstance(points, np.ndarray) and getattr(points, "is_cuda", False):
            return self._call_device(res, info)
        res["mode"] = self.mode
        labeled = self.mode == "train" and res["labeled"]
        if labeled:
            anno = res["lidar"]["annotations"]
            gt_dict = {"gt_boxes": anno["boxes"], "gt_names": np.array(anno["names"]).reshape(-1)}
            _dict_select(gt_dict, kitti.drop_arrays_by_name(gt_dict["gt_names"], ["DontCare", "ignore"]))
            target = np.array([n in self.class_names for n in gt_dict["gt_names"]], dtype=np.bool_)
            if self.db_sampler:
                pasted = self.db_sampler.sample_all(res["metadata"]["image_prefix"], gt_dict["gt_boxes"], gt_dict["gt_names"],
                                                    res["metadata"]["num_point_features"], False, gt_group_ids=None,
                                                    calib=res["calib"] if "calib" in res else None,
                                                    targeted_class_names=self.class_names)
                if pasted is not None:
                    gt_dict["gt_names"] = np.concatenate([gt_dict["gt_names"], pasted["gt_names"]], axis=0)
                    gt_dict["gt_boxes"] = np.concatenate([gt_dict["gt_boxes"], pasted["gt_boxes"]])
                    target = np.concatenate([target, pasted["gt_masks"]], axis=0)
                    if self.remove_points_after_sample:
                        points = points[~box_np_ops.points_in_rbbox(points, pasted["gt_boxes"]).any(-1)]
                    points = np.concatenate([pasted["points"], points], axis=0)
            prep.noise_per_object_v4_(gt_dict["gt_boxes"], points, target, rotation_perturb=self.gt_rotation_noise,
                                      center_noise_std=self.gt_loc_noise_std, global_random_rot_range=self.global_random_rot_range,
                                      group_ids=None, num_try=100, data_aug_with_context=self.data_aug_with_context,
                                      data_aug_random_drop=self.data_aug_random_drop)
            _dict_select(gt_dict, target)
            gt_dict["gt_classes"] = np.array([self.class_names.index(n) + 1 for n in gt_dict["gt_names"]], dtype=np.int32)
            res["lidar"]["points_raw"] = points.copy()
            res["lidar"]["annotations_raw"] = {k: v.copy() for k, v in gt_dict.items()}
            gt_dict["gt_boxes"], points, res["lidar"]["transformation"] = self._global(gt_dict["gt_boxes"], points)
            points = sa_da_v2.pyramid_augment_v0(gt_dict["gt_boxes"], points, **self.sa_da)
        if self.shuffle_points:
            points = points[np.random.choice(np.arange(points.shape[0]), points.shape[0], replace=False)]
        if self.mode == "train" and not res["labeled"]:
            _, points, res["lidar"]["transformation"] = self._global(None, points)
        res["lidar"]["points"] = points
        if labeled:
            res["lidar"]["annotations"] = gt_dict
        return res, info


@PIPELINES.register_module
class Voxelization(object):
    def __init__(self, **kwargs):
        cfg = kwargs.get("cfg", None)
        self.range = _get(cfg, "range")
        self.voxel_size = _get(cfg, "voxel_size")
        self.max_points_in_voxel = _get(cfg, "max_points_in_voxel")
        self.max_voxel_num = _get(cfg, "max_voxel_num")
        self.far_points_first = _get(cfg, "far_points_first", False)
        self.shuffle = False
        self.voxel_generator = VoxelGenerator(point_cloud_range=self.range, voxel_size=self.voxel_size,
                                              max_num_points=self.max_points_in_voxel, max_voxels=self.max_voxel_num)

    def _voxelize(self, points, grid_size):
        voxels, coordinates, num_points_per_voxel = self.voxel_generator.generate(points)
        return dict(voxels=voxels, coordinates=coordinates, num_points=num_points_per_voxel,
                    num_voxels=np.array([voxels.shape[0]], dtype=np.int64), shape=grid_size)

    def __call__(self, res, info):
        grid_size = self.voxel_generator.grid_size
        if res.get("mode") == "train" and res.get("labeled", True) and "annotations" in res["lidar"]:
            # ground truth with no BEV corner inside the x/y range is dropped before target assignment (:200-206)
            from det3d.core.sampler import preprocess as prep
            gt_dict = res["lidar"]["annotations"]
            r = np.asarray(self.voxel_generator.point_cloud_range)
            _dict_select(gt_dict, prep.filter_gt_box_outside_range(gt_dict["gt_boxes"], r[[0, 1, 3, 4]]))
            self.shuffle = True
        res["lidar"]["voxels"] = self._voxelize(res["lidar"]["points"], grid_size)
        if "points_raw" in res["lidar"].keys():
            res["lidar"]["voxels_raw"] = self._voxelize(res["lidar"]["points_raw"], grid_size)
        return res, info


@PIPELINES.register_module
class AssignTarget(object):
    """mirrors det3d/datasets/pipelines/preprocess.py:236-358 for the single-task car configuration: anchors of the one
    `anchor_generator_range` of cfg.target_assigner on the hard-coded [1, 200, 176] feature map, ground-truth boxes of the
    target classes (yaw folded into [-pi, pi)), and per sample the assignment of create_target_np -- run on the MI355X
    (sessd_assign_targets) instead of numpy / numba in a DataLoader worker. Same result layout:
    res["lidar"]["targets"] = {anchors: [(70400,7)], labels: [(70400,)], reg_targets: [(70400,7)], reg_weights: [(70400,)],
    positive_gt_id: [[(P,)]]} and the `targets_raw` twin from `annotations_raw`."""

    def __init__(self, **kwargs):
        import torch
        from sessd_hip.anchors import create_anchors_3d_range
        cfg = kwargs["cfg"]
        ta = cfg["target_assigner"]
        gens = ta["anchor_generators"]
        assert len(gens) == 1, "single anchor generator (config.py:84-94)"
        g = gens[0]
        self.target_class_names = [g["class_name"]]
        self.enable_similar_type = _get(cfg, "enable_similar_type", False)
        self.target_class_ids = [1, 2] if self.enable_similar_type else [1]
        self.matched, self.unmatched = float(g["matched_threshold"]), float(g["unmatched_threshold"])
        self.out_size_factor = _get(cfg, "out_size_factor", 8)
        self.anchors = create_anchors_3d_range((1, 200, 176), g["anchor_ranges"], g["sizes"], g["rotations"]).reshape(-1, 7)
        self._dev_anchors = None
        self._torch = torch

    def _assign(self, gt_boxes, gt_names):
        from sessd_hip import ops
        torch = self._torch
        if self._dev_anchors is None:
            self._dev_anchors = torch.from_numpy(self.anchors).to(torch.device("cuda", torch.cuda.current_device()))
        # assign_v2 :89-95: boxes named like the anchor class (or all of them with enable_similar_type), class id 1
        keep = np.ones(len(gt_names), bool) if self.enable_similar_type else np.array([n == self.target_class_names[0] for n in gt_names], bool)
        g = np.ascontiguousarray(gt_boxes[keep], np.float32).reshape(-1, 7)
        r = ops.assign_targets(self._dev_anchors, torch.from_numpy(g).to(self._dev_anchors.device), None, self.matched, self.unmatched)
        gid = r["gt_id"].cpu().numpy()
        return dict(labels=r["labels"].cpu().numpy(), bbox_targets=r["bbox_targets"].cpu().numpy(),
                    bbox_outside_weights=r["bbox_outside_weights"].cpu().numpy(), positive_gt_id=[gid[gid >= 0]])

    def _targets_of(self, gt_dict):
        mask = np.isin(gt_dict["gt_classes"], self.target_class_ids)
        boxes = gt_dict["gt_boxes"][mask]
        boxes[:, -1] = boxes[:, -1] - np.floor(boxes[:, -1] / (2 * np.pi) + 0.5) * (2 * np.pi)  # limit_period(ry, 0.5, 2 pi)
        gt_dict["gt_boxes"], gt_dict["gt_classes"], gt_dict["gt_names"] = [boxes], [gt_dict["gt_classes"][mask]], [gt_dict["gt_names"][mask]]
        t = self._assign(boxes, gt_dict["gt_names"][0])
        return {"labels": [t["labels"]], "reg_targets": [t["bbox_targets"]], "reg_weights": [t["bbox_outside_weights"]],
                "positive_gt_id": [t["positive_gt_id"]]}

    def __call__(self, res, info):
        targets = {"anchors": [self.anchors]}
        targets_raw = {"anchors": [self.anchors]}
        if res["mode"] == "train" and res.get("labeled", True):
            targets.update(self._targets_of(res["lidar"]["annotations"]))
            if "annotations_raw" in res["lidar"]:
                targets_raw.update(self._targets_of(res["lidar"]["annotations_raw"]))
        res["lidar"]["targets"] = targets
        res["lidar"]["targets_raw"] = targets_raw
        return res, info
