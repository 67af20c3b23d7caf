"""Batch layout VoxelNet.forward expects -- the contract of det3d/torchie/parallel/collate.py:154-218 (collate_kitti) and
det3d/torchie/apis/train_sessd.py:88-106 (example_to_device), as a TABLE: every key of a sample dict belongs to one merge
rule; a key's teacher twin ("<key>_raw", the pre-augmentation copy of formating.py:35-42) follows the same rule.

  rule        keys                                                  batch form
  concat      voxels num_points num_gt voxel_labels num_voxels      rows of all samples under each other (torch)
  batch_index coordinates points                                    same, with the sample index prepended as column 0
  per_task    anchors anchors_mask reg_targets reg_weights labels   list over tasks of (B, ...) stacks (torch)
  pad_boxes   gt_boxes                                              list over tasks of (B, longest, 7) zero-padded float64
  calib       calib                                                 dict of (B, ...) stacks (torch)
  listed      metadata                                              the plain list
  (default)   anything else                                         np.stack
"""
import numpy as np
import torch


def _concat(vals):
    return torch.tensor(np.concatenate(vals, axis=0))


def _batch_index(vals):
    return torch.tensor(np.concatenate([np.pad(v, ((0, 0), (1, 0)), mode="constant", constant_values=b) for b, v in enumerate(vals)], axis=0))


def _per_task(vals):
    return [torch.stack([torch.tensor(sample[t]) for sample in vals]) for t in range(len(vals[0]))]


def _pad_boxes(vals):
    out = []
    for t in range(len(vals[0])):
        padded = np.zeros((len(vals), max(len(sample[t]) for sample in vals), 7))
        for b, sample in enumerate(vals):
            padded[b, :len(sample[t])] = sample[t]
        out.append(padded)
    return out


def _calib(vals):
    names = []
    for sample in vals:
        names += [k for k in sample if k not in names]
    return {k: torch.tensor(np.stack([sample[k] for sample in vals if k in sample], axis=0)) for k in names}


_RULES = {}
for _fn, _keys, _twins in ((_concat, ("voxels", "num_points", "num_gt", "voxel_labels", "num_voxels"), True),
                           (_batch_index, ("coordinates", "points"), True),
                           (_per_task, ("anchors", "anchors_mask", "reg_targets", "reg_weights", "labels"), True),
                           (_pad_boxes, ("gt_boxes",), False), (_calib, ("calib",), False), (list, ("metadata",), False)):
    for _k in _keys:
        _RULES[_k] = _fn
        if _twins:
            _RULES[_k + "_raw"] = _fn
PER_TASK_KEYS = tuple(k for k, f in _RULES.items() if f is _per_task)
TENSOR_KEYS = tuple(k for k, f in _RULES.items() if f in (_concat, _batch_index)) + ("bev_map",)


def collate_kitti(batch_list, samples_per_gpu=1):
    keys = []
    for sample in batch_list:
        keys += [k for k in sample if k not in keys]
    return {k: _RULES.get(k, lambda v: np.stack(v, axis=0))([s[k] for s in batch_list if k in s]) for k in keys}


def example_to_device(example, device=None, non_blocking=False):
    """train_sessd.py:88-106 and the trainer's variant trainer_sessd.py:20-38 (which also moves the teacher's `*_raw` entries):
    tensors and per-task tensor lists go to the device, calib entries become device tensors, the rest passes through."""
    assert device is not None
    move = lambda t: t.to(device, non_blocking=non_blocking)
    out = {}
    for k, v in example.items():
        if k in PER_TASK_KEYS:
            out[k] = [move(t) for t in v]
        elif k in TENSOR_KEYS:
            out[k] = move(v)
        elif k == "calib":
            out[k] = {name: move(torch.as_tensor(t)) for name, t in v.items()}
        else:
            out[k] = v
    return out
