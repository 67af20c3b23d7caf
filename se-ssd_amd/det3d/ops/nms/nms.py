"""mirrors the pybind11 module `det3d.ops.nms.nms` (det3d/ops/nms/nms.cc:3-29: nms_kernel.cu.cc + nms_cpu.h), the
surface det3d/ops/nms/nms_cpu.py:9-27 and nms_gpu.py:172-180 import. Same call signatures (numpy in, Python list /
int out); the work runs on the HIP kernels of libsessd_hip.so -- there is no host implementation behind these names.

  non_max_suppression(boxes, keep_out, thresh, device_id) -> int            nms_kernel.cu.cc (+1 pixel convention)
  non_max_suppression_cpu(boxes, order, thresh, eps=0) -> list[int]         nms_cpu.h:24-70
  rotate_non_max_suppression_cpu(box_corners, order, standup_iou, thresh)   nms_cpu.h:72-168
  IOU_weighted_rotate_non_max_suppression_cpu(... 14 args) -> 5 lists       nms_cpu.h:173-384 (DI-NMS, sessd_di_nms)
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


def IOU_weighted_rotate_non_max_suppression_cpu(boxes, box_corners, standup_iou, thresh, scores, IOU_preds, labels, dirs, anchors,
                                                cnt_thresh, nms_sigma_dist_interval, nms_sigma_square, suppressed_thresh, centerness_c):
    """nms.cc:19-29 / nms_cpu.h:173-384 (DI-NMS core), same 14 arguments and the same [boxes, scores, labels, dirs, keep] list return;
    numpy in, the selection runs on the device (sessd_di_nms). `thresh` is accepted and unused, as in the reference."""
    boxes = np.asarray(boxes, np.float32).reshape(-1, 7)
    n = boxes.shape[0]
    if n == 0:
        return [[], [], [], [], []]
    dev = _dev()
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a), dt)).to(dev)
    an = t(np.asarray(anchors).reshape(n, -1), np.float32) if int(centerness_c) == 1 else None
    b, s, l, d, k = ops.di_nms(t(boxes, np.float32), t(np.asarray(box_corners).reshape(n, 4, 2), np.float32),
                               t(np.asarray(standup_iou).reshape(n, n), np.float32), t(scores, np.float32), t(IOU_preds, np.float32),
                               t(labels, np.int32), t(dirs, np.int32), an, cnt_thresh, nms_sigma_dist_interval, nms_sigma_square,
                               suppressed_thresh)
    return [b.cpu().numpy().tolist(), s.cpu().numpy().tolist(), l.cpu().numpy().tolist(), d.cpu().numpy().tolist(),
            k.cpu().numpy().tolist()]
