"""mirrors det3d/datasets/pipelines/preprocess.py:178-232 (Voxelization): the pipeline stage that feeds the hot path.
Same result dict (`res["lidar"]["voxels"]` = voxels / coordinates / num_points / num_voxels / shape, plus the `*_raw`
twin when `points_raw` is present), voxelized on the MI355X through VoxelGenerator.generate -> sessd_voxelize_frame.
The training-only ground-truth range filter of the reference (:200-206) belongs to data augmentation and is applied
only if the caller provides `filter_gt_box_outside_range` (out of scope here)."""
import numpy as np

from det3d.core.input.voxel_generator import VoxelGenerator

from ..registry import PIPELINES


def _get(cfg, key, default=None):
    return cfg[key] if key in cfg else default


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
