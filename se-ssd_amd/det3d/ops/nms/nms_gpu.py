"""mirrors det3d/ops/nms/nms_gpu.py host functions (numpy in / numpy out like the numba-CUDA originals):
nms_gpu (:132-169), rotate_nms_gpu (:461-499), rotate_iou_gpu (:541-577), rotate_iou_gpu_eval (:636-672)."""
import numpy as np
import torch

from sessd_hip import ops


def _dev(device_id):
    return torch.device("cuda", device_id)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    box_dtype = boxes.dtype
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=np.float32)
    b = torch.from_numpy(np.ascontiguousarray(boxes, np.float32)).to(_dev(device_id))
    q = torch.from_numpy(np.ascontiguousarray(query_boxes, np.float32)).to(_dev(device_id))
    return ops.rotate_iou_eval(b, q, criterion).cpu().numpy().astype(box_dtype)


def rotate_iou_gpu(boxes, query_boxes, device_id=0):
    return rotate_iou_gpu_eval(boxes, query_boxes, -1, device_id)


def _nms(mode, dets, width, thresh, device_id):
    dets = dets.astype(np.float32)
    if dets.shape[0] == 0:
        return []
    order = dets[:, width].argsort()[::-1].astype(np.int32)
    boxes = torch.from_numpy(np.ascontiguousarray(dets[order, :5])).to(_dev(device_id))
    keep, num = ops.nms_sorted(mode, boxes, thresh)
    k = keep[: int(num.item())].cpu().numpy()
    return list(order[k])


def rotate_nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """dets (N,6) [cx,cy,w,l,angle,score]."""
    return _nms(3, dets, 5, nms_overlap_thresh, device_id)


def nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """dets (N,5) [x1,y1,x2,y2,score], +1 pixel convention."""
    return _nms(4, dets, 4, nms_overlap_thresh, device_id)
