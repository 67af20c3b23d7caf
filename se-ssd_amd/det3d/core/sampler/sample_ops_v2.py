"""mirrors det3d/core/sampler/sample_ops_v2.py (DataBaseSamplerV2, the "GT-AUG" sampler built by builder.build_dbsampler):
pastes ground-truth objects of other frames into the current one. Objects come from a per-class database of
dict(name, path, box3d_lidar, num_points_in_gt, difficulty, ...) whose point files hold coordinates relative to the box centre.
Random draws (database shuffles, optional point drop) use numpy's global generator in the reference's order."""
import copy
import pathlib

import numpy as np

from det3d.core.bbox import box_np_ops
from det3d.core.sampler import preprocess as prep


class DataBaseSamplerV2(object):
    def __init__(self, db_infos, groups, db_prepor=None, rate=1.0, global_rot_range=None, logger=None, gt_random_drop=-1.0,
                 gt_aug_with_context=-1.0, gt_aug_similar_type=False):
        if db_prepor is not None:
            db_infos = db_prepor(db_infos)
        self.db_infos, self._rate, self._groups = db_infos, rate, groups
        self._sample_classes = [k for g in groups for k in g.keys()]
        self._sample_max_nums = [v for g in groups for v in g.values()]
        self.gt_point_random_drop, self.gt_aug_with_context = gt_random_drop, gt_aug_with_context
        self._sampler_dict = {k: prep.BatchSampler(v, k) for k, v in db_infos.items()}
        if gt_aug_similar_type:   # vans are pasted as cars
            self._sampler_dict["Car"] = prep.BatchSampler(db_infos["Car"] + db_infos["Van"], "Car")

    def sample_class_v2(self, name, num, gt_boxes):
        """`num` database objects of class `name`, minus those whose BEV footprint (widened by the context margin) collides
        with a present box or with an earlier accepted object (sample_ops_v2.py:198-243). A rejected object does not block."""
        sampled = copy.deepcopy(self._sampler_dict[name].sample(num))
        n_gt = gt_boxes.shape[0]
        sp = np.stack([s["box3d_lidar"] for s in sampled], axis=0)
        ctx = self.gt_aug_with_context if self.gt_aug_with_context > 0.0 else 0.0
        bev = np.concatenate([box_np_ops.center_to_corner_box2d(gt_boxes[:, 0:2], gt_boxes[:, 3:5], gt_boxes[:, -1]),
                              box_np_ops.center_to_corner_box2d(sp[:, 0:2], sp[:, 3:5] + [ctx, ctx], sp[:, -1])], axis=0)
        hit = prep.box_collision_test(bev, bev)
        np.fill_diagonal(hit, False)
        kept = []
        for i in range(n_gt, n_gt + len(sampled)):
            if hit[i].any():
                hit[i], hit[:, i] = False, False
            else:
                kept.append(sampled[i - n_gt])
        return kept

    def sample_all(self, root_path, gt_boxes, gt_names, num_point_features, random_crop=False, gt_group_ids=None, calib=None,
                   targeted_class_names=None, with_road_plane_cam=None):
        """Fills every sample group up to its maximum count (two attempts per class) and returns
        dict(gt_names, difficulty, gt_boxes, points, gt_masks, group_ids) of the pasted objects, or None
        (sample_ops_v2.py:62-196). Frustum cropping of pasted objects (random_crop) is not part of the SE-SSD configuration."""
        if random_crop:
            raise NotImplementedError("random_crop is disabled in the SE-SSD pipeline (preprocess.py:88)")
        want = []
        for cls, max_num in zip(self._sample_classes, self._sample_max_nums):
            missing = int(max_num - np.sum([n == cls for n in gt_names]))
            want.append(np.round(self._rate * missing).astype(np.int64))
        sampled, sampled_boxes, present = [], [], gt_boxes
        for cls, num in zip(self._sample_classes, want):
            tries = 0
            while num > 0 and tries < 2:
                objs = self.sample_class_v2(cls, num, present)
                sampled += objs
                if objs:
                    b = np.stack([o["box3d_lidar"] for o in objs], axis=0)
                    sampled_boxes.append(b)
                    present = np.concatenate([present, b], axis=0)
                num -= len(objs)
                tries += 1
        if not sampled:
            return None
        sampled_boxes = np.concatenate(sampled_boxes, axis=0)
        clouds = []
        for k, info in enumerate(sampled):
            pts = np.fromfile(str(pathlib.Path(root_path) / info["path"]), dtype=np.float32).reshape(-1, num_point_features)
            pts[:, :3] += info["box3d_lidar"][:3]
            if with_road_plane_cam is not None:   # put the object on the frame's road plane a x + b y + c z + d = 0 (camera frame)
                a, b, c, d = with_road_plane_cam
                cam = info["box3d_cam"]
                lift = cam[1] - (-d - a * cam[0] - c * cam[2]) / b
                pts[:, 2] += lift
                sampled_boxes[sampled.index(info), 2] += lift
            if self.gt_point_random_drop > 0 and pts.shape[0] > 10:
                drop = int(np.random.uniform(0, self.gt_point_random_drop) * pts.shape[0])
                pts = pts[np.random.choice(np.arange(pts.shape[0]), pts.shape[0] - drop, replace=False)]
            clouds.append(pts)
        return {"gt_names": np.array([s["name"] for s in sampled]), "difficulty": np.array([s["difficulty"] for s in sampled]),
                "gt_boxes": sampled_boxes, "points": np.concatenate(clouds, axis=0),
                "gt_masks": np.ones((len(sampled),), dtype=np.bool_),
                "group_ids": np.arange(gt_boxes.shape[0], gt_boxes.shape[0] + len(sampled))}
