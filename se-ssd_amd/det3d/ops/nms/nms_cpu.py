"""mirrors det3d/ops/nms/nms_cpu.py:34-127: the numpy-level NMS helpers around the `det3d.ops.nms.nms` module.
Same names and return conventions (lists of kept indices into the input)."""
import numpy as np

from det3d.core.bbox import box_np_ops
from det3d.ops.nms.nms import (IOU_weighted_rotate_non_max_suppression_cpu, non_max_suppression_cpu,  # noqa: F401
                               rotate_non_max_suppression_cpu)


def nms_cc(dets, thresh):
    """dets (N,5) [x1,y1,x2,y2,score] (nms_cpu.py:34-37: eps = 1)."""
    order = dets[:, 4].argsort()[::-1].astype(np.int32)
    return non_max_suppression_cpu(dets, order, thresh, 1.0)


def rotate_nms_cc(dets, thresh):
    """dets (N,6) [x,y,w,l,r,score] (nms_cpu.py:40-51)."""
    order = dets[:, 5].argsort()[::-1].astype(np.int32)
    corners = box_np_ops.center_to_corner_box2d(dets[:, :2], dets[:, 2:4], dets[:, 4])
    standup = box_np_ops.corner_to_standup_nd(corners)
    standup_iou = box_np_ops.iou_jit(standup, standup, eps=0.0)
    return rotate_non_max_suppression_cpu(corners, order, standup_iou, thresh)


def nms_jit(dets, thresh, eps=0.0):
    """nms_cpu.py:100-127: same greedy loop as non_max_suppression_cpu with the order taken from column 4."""
    order = dets[:, 4].argsort()[::-1].astype(np.int32)
    return non_max_suppression_cpu(dets, order, thresh, eps)


def rotate_weighted_nms_cc(box, dets, thresh, iou_preds, labels, dirs, anchors=None, nms_cnt_thresh=2.6,
                           nms_sigma_dist_interval=(0, 20, 40, 60), nms_sigma_square=(0.0009, 0.009, 0.1, 1), suppressed_thresh=0.3):
    """DI-NMS on numpy inputs (nms_cpu.py:52-93): box (N,7) predictions, dets (N,6) [x,y,w,l,r,score]; anchors given => centerness
    damping inside the core. Returns the core's [boxes, scores, labels, dirs, keep] lists."""
    scores = dets[:, 5]
    corners = box_np_ops.center_to_corner_box2d(dets[:, :2], dets[:, 2:4], dets[:, 4])
    standup = box_np_ops.corner_to_standup_nd(corners)
    standup_iou = box_np_ops.iou_jit(standup, standup, eps=0.0)
    if anchors is None:
        centerness_c, anchors = 0, np.zeros((1, 1))
    else:
        centerness_c = 1
    return IOU_weighted_rotate_non_max_suppression_cpu(box, corners, standup_iou, thresh, scores, iou_preds, labels, dirs, anchors,
                                                       nms_cnt_thresh, nms_sigma_dist_interval, nms_sigma_square, suppressed_thresh,
                                                       centerness_c)
