"""mirrors det3d/datasets/kitti/kitti.py (KittiDataset): the per-frame entry of the data pipeline (`get_sensor_data` :170-216:
the `res` dict every pipeline stage reads and writes), detections of the model (lidar boxes) -> KITTI annotation dicts in the
rectified camera frame with 2-D boxes by projection (:71-139), and `evaluation` (:141-166) = that conversion + the official
average precision. The infos come from `info_path` (the reference's kitti_infos_*.pkl) or are handed in as `kitti_infos`."""
import os
import pickle

import numpy as np

from det3d.core.bbox import box_np_ops
from det3d.datasets.kitti.eval import get_coco_eval_result, get_official_eval_result, get_official_eval_result_v2

_KEYS = ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score")


def empty_result_anno():
    a = {k: np.array([]) for k in _KEYS}
    a.update(bbox=np.zeros([0, 4]), dimensions=np.zeros([0, 3]), location=np.zeros([0, 3]))
    return a


def convert_detection_to_kitti_annos(detection, kitti_infos, class_names, partial=False):
    """detection: {token: dict(box3d_lidar (n,7) [x,y,z,w,l,h,r], scores, label_preds, metadata)} (torch or numpy);
    kitti_infos: list of dict(image=dict(image_idx, image_shape (h, w)), calib=dict(R0_rect, Tr_velo_to_cam, P2)).
    One annotation dict per info (per detection key with partial=True). Boxes whose projection lies outside the image are dropped,
    the rest clipped to it; alpha = -atan2(-y, x) + ry; yaw folded into [-pi, pi); z moved from the centre to the bottom face."""
    # copies: the yaw fold / bottom-face shift below must not write through to the caller's detections
    to_np = lambda v: v.detach().cpu().numpy().copy() if hasattr(v, "detach") else np.array(v)
    gt_ids = [str(info["image"]["image_idx"]) for info in kitti_infos]
    annos = []
    for key in (list(detection.keys()) if partial else gt_ids):
        det = detection[key]
        info = kitti_infos[gt_ids.index(key)]
        calib = info["calib"]
        boxes, labels, scores = to_np(det["box3d_lidar"]), to_np(det["label_preds"]), to_np(det["scores"])
        rows = {k: [] for k in _KEYS}
        if boxes.shape[0] != 0:
            boxes[:, -1] = box_np_ops.limit_period(boxes[:, -1], offset=0.5, period=np.pi * 2)
            boxes[:, 2] -= boxes[:, 5] / 2
            cam = box_np_ops.box_lidar_to_camera(boxes, calib["R0_rect"], calib["Tr_velo_to_cam"])
            corners = box_np_ops.center_to_corner_box3d(cam[:, :3], cam[:, 3:6], cam[:, 6], [0.5, 1.0, 0.5], axis=1)
            uv = box_np_ops.project_to_image(corners, calib["P2"])
            bbox = np.concatenate([uv.min(axis=1), uv.max(axis=1)], axis=1)
            h, w = info["image"]["image_shape"][0], info["image"]["image_shape"][1]
            for j in range(cam.shape[0]):
                if bbox[j, 0] > w or bbox[j, 1] > h or bbox[j, 2] < 0 or bbox[j, 3] < 0:
                    continue
                bbox[j, 2:] = np.minimum(bbox[j, 2:], [w, h])
                bbox[j, :2] = np.maximum(bbox[j, :2], [0, 0])
                rows["bbox"].append(bbox[j])
                rows["alpha"].append(-np.arctan2(-boxes[j, 1], boxes[j, 0]) + cam[j, 6])
                rows["dimensions"].append(cam[j, 3:6])
                rows["location"].append(cam[j, :3])
                rows["rotation_y"].append(cam[j, 6])
                rows["name"].append(class_names[int(labels[j])])
                rows["truncated"].append(0.0)
                rows["occluded"].append(0)
                rows["score"].append(scores[j])
        anno = {k: np.stack(v) for k, v in rows.items()} if rows["name"] else empty_result_anno()
        anno["metadata"] = det["metadata"]
        annos.append(anno)
    return annos


class KittiDataset(object):
    NumPointFeatures = 4

    def __init__(self, root_path=None, info_path=None, cfg=None, pipeline=None, class_names=None, test_mode=False,
                 kitti_infos=None, **kwargs):
        if kitti_infos is None:
            assert info_path is not None
            with open(info_path, "rb") as f:
                kitti_infos = pickle.load(f)
        self._kitti_infos, self._class_names = kitti_infos, class_names
        self._root_path, self._info_path, self.test_mode = root_path, info_path, test_mode
        self.labeled = kwargs.get("labeled", True)
        self.plane_dir = None if root_path is None else str(root_path) + "/training/planes"
        if pipeline is None or callable(pipeline):
            self.pipeline = pipeline
        else:
            from det3d.datasets.pipelines import Compose
            self.pipeline = Compose(pipeline)

    def __len__(self):
        return len(self._kitti_infos)

    num_point_features = property(lambda self: KittiDataset.NumPointFeatures)

    def get_road_plane(self, idx):
        """unit normal (facing up in the rectified camera frame) and offset of the frame's road plane, 4th line of planes/%06d.txt (:43-57)"""
        with open(os.path.join(self.plane_dir, "%06d.txt" % idx), "r") as f:
            plane = np.asarray([float(v) for v in f.readlines()[3].split()])
        if plane[1] > 0:
            plane = -plane
        return plane / np.linalg.norm(plane[0:3])

    def get_sensor_data(self, idx, with_image=False, with_gp=False, by_index=False):
        """the frame's `res` dict run through the pipeline (:170-216). by_index: idx is an image index, not a position."""
        if by_index:
            idx = [int(i["image"]["image_idx"]) for i in self._kitti_infos].index(idx)
        info = self._kitti_infos[idx]
        if with_image:
            raise NotImplementedError("camera images are not part of the lidar path")
        res = {"type": "KittiDataset",
               "lidar": {"type": "lidar", "points": None, "ground_plane": -self.get_road_plane(idx)[-1] if with_gp else None,
                         "annotations": None, "names": None, "targets": None},
               "metadata": {"image_prefix": self._root_path, "num_point_features": KittiDataset.NumPointFeatures,
                            "image_idx": info["image"]["image_idx"], "image_shape": info["image"]["image_shape"],
                            "token": str(info["image"]["image_idx"])},
               "calib": None, "cam": {"annotations": None}, "mode": "val" if self.test_mode else "train", "labeled": self.labeled}
        data, _ = self.pipeline(res, info)
        return data

    def __getitem__(self, idx):
        return self.get_sensor_data(idx, with_gp=False)

    @property
    def ground_truth_annotations(self):
        return None if "annos" not in self._kitti_infos[0] else [info["annos"] for info in self._kitti_infos]

    def convert_detection_to_kitti_annos(self, detection, partial=False):
        return convert_detection_to_kitti_annos(detection, self._kitti_infos, self._class_names, partial)

    def evaluation(self, detections, output_dir=None, get_results=True):
        """KITTI camera boxes: height axis 1, location at the bottom face (z_center 1.0) (:141-166)."""
        dt_annos = self.convert_detection_to_kitti_annos(detections)
        results = None
        if get_results:
            gt = self.ground_truth_annotations
            official = get_official_eval_result(gt, dt_annos, self._class_names, z_axis=1, z_center=1.0)
            coco = get_coco_eval_result(gt, dt_annos, self._class_names, z_axis=1, z_center=1.0)
            r40 = get_official_eval_result_v2(gt, dt_annos, self._class_names, z_axis=1, z_center=1.0)
            results = {"results": {"official_AP_11": official["result"]}, "results_2": {"official_AP_40": r40["result"]},
                       "detail": {"eval.kitti": {"official": official["detail"], "coco": coco["detail"]}}}
        return results, dt_annos
