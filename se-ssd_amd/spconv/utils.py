"""spconv v1's host helpers `spconv.utils.rbbox_iou` / `rbbox_intersection`, imported by det3d/core/bbox/box_np_ops.py:9 and
used by riou_cc / rinter_cc (:20-50): numpy in, numpy out, computed by the device quad clipper (sessd_quads_pairwise).
PARITY UNPINNED like the rest of spconv (third-party, absent): semantics restated -- pairwise IoU / intersection area of convex
quads given as corners (N,4,2) x (K,4,2); pairs whose stand-up IoU is <= standup_thresh stay 0."""
import numpy as np
import torch

from sessd_hip import ops


def _run(mode, box_corners, qbox_corners, standup_iou, standup_thresh):
    box_corners, qbox_corners = np.asarray(box_corners), np.asarray(qbox_corners)
    if box_corners.shape[0] == 0 or qbox_corners.shape[0] == 0:
        return np.zeros((box_corners.shape[0], qbox_corners.shape[0]), box_corners.dtype)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    out = ops.quads_pairwise(mode, t(box_corners), t(qbox_corners), t(standup_iou), float(standup_thresh))
    return out.cpu().numpy().astype(box_corners.dtype)


def rbbox_iou(box_corners, qbox_corners, standup_iou, standup_thresh):
    return _run(0, box_corners, qbox_corners, standup_iou, standup_thresh)


def rbbox_intersection(box_corners, qbox_corners, standup_iou, standup_thresh):
    return _run(1, box_corners, qbox_corners, standup_iou, standup_thresh)
