"""the two annotation filters of det3d/datasets/kitti/kitti_common.py that the data pipeline calls (:506-511, :550-553)."""
import numpy as np


def remove_dontcare(image_anno):
    """annotation dict without the rows named "DontCare"."""
    keep = [i for i, x in enumerate(image_anno["name"]) if x != "DontCare"]
    return {k: v[keep] for k, v in image_anno.items()}


def drop_arrays_by_name(gt_names, used_classes):
    """indices of the names NOT in used_classes."""
    return np.array([i for i, x in enumerate(gt_names) if x not in used_classes], dtype=np.int64)
