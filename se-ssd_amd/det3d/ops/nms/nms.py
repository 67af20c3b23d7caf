"""mirrors the pybind11 module `det3d.ops.nms.nms` (det3d/ops/nms/nms.cc:3-29: nms_kernel.cu.cc + nms_cpu.h), the
surface det3d/ops/nms/nms_cpu.py:9-27 and nms_gpu.py:172-180 import. Same call signatures (numpy in, Python list /
int out); the work runs on the HIP kernels of libsessd_hip.so -- there is no host implementation behind these names.

  non_max_suppression(boxes, keep_out, thresh, device_id) -> int            nms_kernel.cu.cc (+1 pixel convention)
  non_max_suppression_cpu(boxes, order, thresh, eps=0) -> list[int]         nms_cpu.h:24-70
  rotate_non_max_suppression_cpu(box_corners, order, standup_iou, thresh)   nms_cpu.h:72-168
  IOU_weighted_rotate_non_max_suppression_cpu(...)                          nms_cpu.h:173-384 (DI-NMS: SURVEY 8f row 4, not built)
"""
import numpy as np
import torch

from sessd_hip import ops


def _dev(device_id=0):
    return torch.device("cuda", int(device_id))


def non_max_suppression(boxes, keep_out, thresh, device_id=0):
    """boxes (N,5) [x1,y1,x2,y2,score] ALREADY sorted by descending score (nms_gpu.py:172-180 sorts before the call);
    writes the kept row numbers into keep_out and returns their count."""
    b = torch.from_numpy(np.ascontiguousarray(boxes, np.float32)).to(_dev(device_id))
    keep, num = ops.nms_sorted(4, b, thresh)
    n = int(num.item())
    keep_out[:n] = keep[:n].cpu().numpy()
    return n


def non_max_suppression_cpu(boxes, order, thresh, eps=0.0):
    boxes = np.ascontiguousarray(boxes, np.float32)
    order = np.asarray(order, np.int64)
    if boxes.shape[0] == 0:
        return []
    b = torch.from_numpy(np.ascontiguousarray(boxes[order])).to(_dev())
    keep, num = ops.nms_axis_eps_sorted(b, thresh, eps)
    n = int(num.item())
    return [int(v) for v in order[keep[:n].cpu().numpy()]]


def rotate_non_max_suppression_cpu(box_corners, order, standup_iou, thresh):
    """box_corners (K,4,2), order (K,) int32 (descending score), standup_iou (K,K) -- recomputed on the device from the
    corners' bounding boxes, which is what nms_cpu.py:45-49 passes in."""
    corners = np.ascontiguousarray(box_corners, np.float32)
    order = np.asarray(order, np.int64)
    k = corners.shape[0]
    if k == 0:
        return []
    c = torch.from_numpy(np.ascontiguousarray(corners[order])).to(_dev())
    keep, num = ops.rotate_nms_corners_sorted(c, thresh, k)
    n = int(num.item())
    return [int(v) for v in order[keep[:n].cpu().numpy().astype(np.int64)]]


def IOU_weighted_rotate_non_max_suppression_cpu(*args, **kwargs):
    raise NotImplementedError("DI-NMS (nms_cpu.h:173-384) is a 'next' row of the scope table (SURVEY 8f-4) and is not built; "
                              "SE-SSD's config.py test_cfg uses plain rotate_nms")
