"""mirrors det3d/datasets/utils/create_gt_database.py:20-131: the producer of the ground-truth database the GT-AUG sampler reads
(core/sampler/sample_ops_v2.py). For every labelled object of every frame: the points inside its box (optionally inside the box
widened by `gt_aug_with_context` in w and l), stored centre-relative as float32 x,y,z,intensity in
`<root>/gt_database/<image_idx>_<name>_<i>.bin`, and a record dict(name, path, image_idx, gt_idx, box3d_lidar,
num_points_in_gt, difficulty, group_id) in `<root>/dbinfos_train.pkl`, grouped by class name."""
import pickle
from pathlib import Path

import numpy as np

from det3d.core.bbox import box_np_ops


def create_groundtruth_database(dataset_class_name, data_path, info_path=None, used_classes=None, db_path=None, dbinfo_path=None,
                                relative_path=True, add_rgb=False, lidar_only=False, bev_only=False, coors_range=None,
                                gt_aug_with_context=-1.0, **kwargs):
    if dataset_class_name != "KITTI":
        raise NotImplementedError("only the KITTI dataset is on the SE-SSD path")
    from det3d.datasets.kitti.kitti import KittiDataset
    dataset = KittiDataset(info_path=info_path, root_path=data_path, test_mode=True, pipeline=[
        {"type": "LoadPointCloudFromFile", "dataset": "KittiDataset"},
        {"type": "LoadPointCloudAnnotations", "with_bbox": True, "enable_difficulty_level": True}])
    root = Path(data_path)
    widened = gt_aug_with_context > 0.0
    if widened:   # the widened crops live beside the plain ones (builder.build_dbsampler switches to them)
        db_path, dbinfo_path = root / "gt_enlarged_database", root / "dbinfos_enlarged_train.pkl"
    else:
        db_path = root / "gt_database" if db_path is None else Path(db_path)
        dbinfo_path = root / "dbinfos_train.pkl" if dbinfo_path is None else Path(dbinfo_path)
    db_path.mkdir(parents=True, exist_ok=True)
    grow = np.array([0, 0, 0, gt_aug_with_context, gt_aug_with_context, 0, 0] if widened else [0.0] * 7)
    by_class, next_group = {}, 0
    for index in range(len(dataset)):
        frame = dataset.get_sensor_data(index)
        image_idx = frame["metadata"].get("image_idx", index)
        points, annos = frame["lidar"]["points"], frame["lidar"]["annotations"]
        boxes, names = annos["boxes"], annos["names"]
        n = boxes.shape[0]
        group_ids = annos["group_ids"] if "group_ids" in annos else np.arange(n, dtype=np.int64)
        difficulty = annos["difficulty"] if "difficulty" in annos else np.zeros(n, dtype=np.int32)
        inside = box_np_ops.points_in_rbbox(points, boxes)
        crop = box_np_ops.points_in_rbbox(points, boxes + grow) if widened else inside
        frame_groups = {}
        for i in range(n):
            filename = "%s_%s_%d.bin" % (image_idx, names[i], i)
            obj = points[crop[:, i]]
            obj[:, :3] -= boxes[i, :3]
            obj[:, :4].tofile(str(db_path / filename))
            if used_classes is not None and names[i] not in used_classes:
                continue
            if group_ids[i] not in frame_groups:
                frame_groups[group_ids[i]] = next_group
                next_group += 1
            rec = {"name": names[i], "path": db_path.stem + "/" + filename if relative_path else str(db_path / filename),
                   "image_idx": image_idx, "gt_idx": i, "box3d_lidar": boxes[i], "num_points_in_gt": inside[:, i].sum(),
                   "difficulty": difficulty[i], "group_id": frame_groups[group_ids[i]]}
            if "score" in annos:
                rec["score"] = annos["score"][i]
            by_class.setdefault(names[i], []).append(rec)
    with open(dbinfo_path, "wb") as f:
        pickle.dump(by_class, f)
    return by_class
