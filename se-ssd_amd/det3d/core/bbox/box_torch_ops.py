"""mirrors det3d/core/bbox/box_torch_ops.py: second_box_encode/decode (:26-147), nms (:505-524), rotate_nms (:527-548).

rotate_nms keeps the reference signature but never leaves the device: topk on the HIP device, rotated NMS by
libsessd_hip.so (sessd_rotate_nms_sorted), one tiny read of the kept count at the end (the reference does a
D2H + CPU boost NMS + H2D here)."""
import torch

from sessd_hip import ops


def second_box_encode(boxes, anchors, encode_angle_to_vector=False, smooth_dim=False, norm_velo=False):
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xg, yg, zg, wg, lg, hg, rg = torch.split(boxes, 1, dim=-1)
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xt, yt, zt = (xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / ha
    if smooth_dim:
        lt, wt, ht = lg / la - 1, wg / wa - 1, hg / ha - 1
    else:
        lt, wt, ht = torch.log(lg / la), torch.log(wg / wa), torch.log(hg / ha)
    if encode_angle_to_vector:
        return torch.cat([xt, yt, zt, wt, lt, ht, torch.cos(rg) - torch.cos(ra), torch.sin(rg) - torch.sin(ra)], dim=-1)
    return torch.cat([xt, yt, zt, wt, lt, ht, rg - ra], dim=-1)


def second_box_decode(box_encodings, anchors, encode_angle_to_vector=False, bin_loss=False, smooth_dim=False,
                      norm_velo=False):
    """box decode for VoxelNet in lidar: boxes/anchors [...,7] x,y,z,w,l,h,r."""
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    if encode_angle_to_vector:
        xt, yt, zt, wt, lt, ht, rtx, rty = torch.split(box_encodings, 1, dim=-1)
    else:
        xt, yt, zt, wt, lt, ht, rt = torch.split(box_encodings, 1, dim=-1)
    diagonal = torch.sqrt(la ** 2 + wa ** 2)
    xg, yg, zg = xt * diagonal + xa, yt * diagonal + ya, zt * ha + za
    if smooth_dim:
        lg, wg, hg = (lt + 1) * la, (wt + 1) * wa, (ht + 1) * ha
    else:
        lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
    if encode_angle_to_vector:
        rg = torch.atan2(rty + torch.sin(ra), rtx + torch.cos(ra))
    else:
        rg = rt + ra
    return torch.cat([xg, yg, zg, wg, lg, hg, rg], dim=-1)


def rotation_points_single_angle(points, angle, axis=0):
    """points (N,3) @ R^T for one scalar angle about `axis` (box_torch_ops.py:320-345); axis 2: x' = x cos + y sin,
    y' = -x sin + y cos."""
    import math
    s, c = math.sin(angle), math.cos(angle)
    if axis == 1:
        m = [[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]]
    elif axis in (2, -1):
        m = [[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]
    elif axis == 0:
        m = [[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]]
    else:
        raise ValueError("axis should in range")
    return points @ torch.tensor(m, dtype=points.dtype, device=points.device)


def rotate_nms(rbboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """rbboxes (K,5) [x,y,w,l,r], scores (K,) -> LongTensor of kept indices into the input (<= post_max_size)."""
    if rbboxes.shape[0] == 0:
        return torch.zeros([0]).long().to(rbboxes.device)
    if pre_max_size is not None:
        k = min(scores.shape[0], pre_max_size)
        scores, indices = torch.topk(scores, k=k)
        rbboxes = rbboxes[indices]
    else:
        scores, indices = torch.sort(scores, descending=True)
        rbboxes = rbboxes[indices]
    post = int(post_max_size) if post_max_size is not None else int(rbboxes.shape[0])
    keep, num = ops.rotate_nms_sorted(rbboxes.float().contiguous(), iou_threshold, post)
    keep = keep[: int(num.item())].long()
    if keep.shape[0] == 0:
        return torch.zeros([0]).long().to(rbboxes.device)
    return indices[keep]


def rotate_weighted_nms(box_preds, rbboxes, dir_labels, labels_preds, scores, iou_preds, anchors, enable_centerness=True,
                        centerness_pow=1, centerness_c=False, pre_max_size=None, post_max_size=None, iou_threshold=0.5,
                        nms_cnt_thresh=2.6, nms_sigma_dist_interval=(0, 20, 40, 60), nms_sigma_square=(0.0009, 0.009, 0.1, 1),
                        suppressed_thresh=0.3):
    """DI-NMS (box_torch_ops.py:552-621): confidence-aware, IoU-weighted box averaging instead of plain suppression. box_preds
    (K,7), rbboxes (K,5) [x,y,w,l,r], dir_labels / labels_preds (K,), scores / iou_preds (K,), anchors (K,7). Returns (boxes (k,7),
    dirs (k,), labels (k,), scores (k,), kept indices into the input) on the input's device. Like the reference, the input
    `scores` tensor is damped in place when enable_centerness is set without centerness_c, and an empty input returns None.
    (Without pre_max_size the reference fails on an unbound `indices`; here the indices are then simply 0..K-1.)
    LIMIT of the device kernel: at most 1024 boxes after the top-k (config.py: nms_pre_max_size = 1000); pass pre_max_size <= 1024
    -- more raises ValueError (the reference's host loop takes any K)."""
    indices = torch.arange(scores.shape[0], device=scores.device)
    if pre_max_size is not None:
        k = min(scores.shape[0], pre_max_size)
        scores, indices = torch.topk(scores, k=k)
        rbboxes, iou_preds, dir_labels = rbboxes[indices], iou_preds[indices], dir_labels[indices]
        labels_preds, box_preds, anchors = labels_preds[indices], box_preds[indices], anchors[indices]
    if enable_centerness and not centerness_c:
        dist = torch.abs(box_preds - anchors)
        metric = torch.softmax(torch.pow(torch.pow(dist[:, 0:2], 2).sum(-1), 0.5), dim=0)
        scores *= torch.pow(torch.ones_like(metric) - metric, centerness_pow)
    import numpy as np
    from det3d.ops.nms.nms_cpu import rotate_weighted_nms_cc
    if rbboxes.shape[0] == 0:
        return None   # the reference's empty branch falls off the end of the function (box_torch_ops.py:598-599)
    dev = rbboxes.device
    # as in the reference the footprints, stand-up boxes and their IoU are prepared in numpy (nms_cpu.py:65-72); the selection
    # loop itself -- the O(K^2) polygon work -- runs on the device
    dets_np = torch.cat([rbboxes, scores.unsqueeze(-1)], dim=1).data.cpu().numpy()
    res = rotate_weighted_nms_cc(box_preds.data.cpu().numpy(), dets_np, iou_threshold, iou_preds.data.cpu().numpy().reshape(-1),
                                 labels_preds.cpu().numpy().reshape(-1), dir_labels.cpu().numpy().reshape(-1),
                                 anchors.cpu().numpy() if (enable_centerness and centerness_c) else None,
                                 nms_cnt_thresh=nms_cnt_thresh, nms_sigma_dist_interval=nms_sigma_dist_interval,
                                 nms_sigma_square=nms_sigma_square, suppressed_thresh=suppressed_thresh)
    keep = torch.from_numpy(np.array(res[4], dtype=np.int64)).to(dev)
    return (torch.from_numpy(np.array(res[0], dtype=np.float32).reshape(-1, 7)).to(dev), torch.from_numpy(np.array(res[3], dtype=np.int64)).to(dev),
            torch.from_numpy(np.array(res[2], dtype=np.int64)).to(dev), torch.from_numpy(np.array(res[1], dtype=np.float32)).to(dev), indices[keep])


def nms(bboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """axis-aligned NMS (box_torch_ops.py:505-524 -> nms_gpu, numba kernel with the +1 pixel convention); boxes (K,4)."""
    if bboxes.shape[0] == 0:
        return torch.zeros([0]).long().to(bboxes.device)
    k = scores.shape[0] if pre_max_size is None else min(scores.shape[0], pre_max_size)
    scores, indices = torch.topk(scores, k=k)
    b5 = torch.cat([bboxes[indices].float(), torch.zeros((k, 1), device=bboxes.device)], 1).contiguous()
    keep, num = ops.nms_sorted(4, b5, iou_threshold)
    keep = keep[: int(num.item())]
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return indices[keep]
