"""mirrors det3d/torchie/parallel/collate.py:154-218 (collate_kitti): the batch layout VoxelNet.forward expects --
voxels / num_points / num_voxels concatenated over the batch, coordinates left-padded with the batch index,
anchors stacked per task, calib entries stacked, metadata as a list."""
import collections

import numpy as np
import torch


def collate_kitti(batch_list, samples_per_gpu=1):
    example_merged = collections.defaultdict(list)
    for example in batch_list:
        for k, v in example.items():
            example_merged[k].append(v)
    ret = {}
    for key, elems in example_merged.items():
        if key in ["voxels", "num_points", "num_gt", "voxel_labels", "num_voxels", "voxels_raw", "num_points_raw",
                   "num_gt_raw", "voxel_labels_raw", "num_voxels_raw"]:
            ret[key] = torch.tensor(np.concatenate(elems, axis=0))
        elif key == "gt_boxes":  # per task: boxes of every sample zero-padded to the longest list of the batch
            ret[key] = []
            for task in range(len(elems[0])):
                width = max(len(e[task]) for e in elems)
                padded = np.zeros((len(elems), width, 7))
                for b, e in enumerate(elems):
                    padded[b, :len(e[task])] = e[task]
                ret[key].append(padded)
        elif key == "metadata":
            ret[key] = elems
        elif key == "calib":
            ret[key] = {}
            for elem in elems:
                for k1, v1 in elem.items():
                    ret[key].setdefault(k1, []).append(v1)
            for k1, v1 in ret[key].items():
                ret[key][k1] = torch.tensor(np.stack(v1, axis=0))
        elif key in ["coordinates", "points", "coordinates_raw", "points_raw"]:
            coors = [np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, c in enumerate(elems)]
            ret[key] = torch.tensor(np.concatenate(coors, axis=0))
        elif key in ["anchors", "anchors_mask", "reg_targets", "reg_weights", "labels", "anchors_raw", "anchors_mask_raw",
                     "reg_targets_raw", "reg_weights_raw", "labels_raw"]:
            per_task = collections.defaultdict(list)
            for elem in elems:
                for idx, ele in enumerate(elem):
                    per_task[str(idx)].append(torch.tensor(ele))
            ret[key] = [torch.stack(vv) for vv in per_task.values()]
        else:
            ret[key] = np.stack(elems, axis=0)
    return ret


def example_to_device(example, device=None, non_blocking=False):
    """mirrors det3d/torchie/apis/train_sessd.py:88-106 and the trainer's variant det3d/torchie/trainer/trainer_sessd.py:20-38,
    which also moves the teacher's `*_raw` inputs and targets."""
    assert device is not None
    out = {}
    for k, v in example.items():
        if k in ["anchors", "anchors_mask", "reg_targets", "reg_weights", "labels", "anchors_raw", "anchors_mask_raw",
                 "reg_targets_raw", "reg_weights_raw", "labels_raw"]:
            out[k] = [res.to(device, non_blocking=non_blocking) for res in v]
        elif k in ["voxels", "bev_map", "coordinates", "num_points", "points", "num_voxels", "voxels_raw", "coordinates_raw",
                   "num_points_raw", "points_raw", "num_voxels_raw"]:
            out[k] = v.to(device, non_blocking=non_blocking)
        elif k == "calib":
            out[k] = {k1: torch.as_tensor(v1).to(device, non_blocking=non_blocking) for k1, v1 in v.items()}
        else:
            out[k] = v
    return out
