"""mirrors det3d/core/iou3d/iou3d_utils.py (GPU functions :32-52,143-306) and utils.py:74-126 converters."""
import torch

import iou3d_cuda


def boxes3d_to_bev_torch(boxes3d, box_mode="wlh", rect=False):
    """(N,7) [x,y,z,<box_mode>,ry] or (N,5) [x,y,<w,l order>,ry] -> [x1,y1,x2,y2,ry] (utils.py:74-101).
    velodyne (rect=False): centre (x, y), half extents (w/2, l/2); camera (rect=True): centre (x, z), half extents (l/2, w/2)."""
    cols = boxes3d.shape[-1]
    if cols not in (5, 7):
        raise NotImplementedError
    first = 2 if cols == 5 else 3
    half_w, half_l = boxes3d[:, box_mode.index("w") + first] / 2, boxes3d[:, box_mode.index("l") + first] / 2
    cu = boxes3d[:, 0]
    cv, eu, ev = (boxes3d[:, 2], half_l, half_w) if rect else (boxes3d[:, 1], half_w, half_l)
    return torch.stack([cu - eu, cv - ev, cu + eu, cv + ev, boxes3d[:, -1]], dim=1)


def boxes3d_to_bev_3d_torch(boxes3d, box_mode="wlh", rect=False):
    """(N,7) -> [x1,y1,z1,x2,y2,z2,ry] (utils.py:104-126): velodyne z +- h/2; camera: (x, z) footprint with swapped half
    extents and the height interval [y - h, y]."""
    half_w, half_l = boxes3d[:, box_mode.index("w") + 3] / 2, boxes3d[:, box_mode.index("l") + 3] / 2
    h = boxes3d[:, box_mode.index("h") + 3]
    cu = boxes3d[:, 0]
    if rect:
        cv, cw = boxes3d[:, 2], boxes3d[:, 1]
        lo, hi = [cu - half_l, cv - half_w, cw - h], [cu + half_l, cv + half_w, cw]
    else:
        cv, cw = boxes3d[:, 1], boxes3d[:, 2]
        lo, hi = [cu - half_w, cv - half_l, cw - h / 2], [cu + half_w, cv + half_l, cw + h / 2]
    return torch.stack(lo + hi + [boxes3d[:, 6]], dim=1)


def rbbox2d_to_near_bbox_torch(in_boxes, box_mode="wlh", rect=False):
    """Rotated box -> the nearest axis-aligned ('standing' or 'lying') box [x1,y1,x2,y2,0]: sizes swap when the yaw folded
    into [-pi/2, pi/2) is more than 45 degrees off the axis (utils.py:46-72)."""
    import math
    if in_boxes.shape[-1] == 7:
        wi, li = box_mode.index("w") + 3, box_mode.index("l") + 3
        b = in_boxes[:, [0, 2 if rect else 1, wi, li, -1]]
    else:
        wi, li = box_mode.index("w") + 2, box_mode.index("l") + 2
        b = in_boxes[:, [0, 1, wi, li, -1]]
    rot = b[:, 4]
    folded = torch.abs(rot - torch.floor(rot / math.pi + 0.5) * math.pi)  # limit_period(rot, 0.5, pi)
    swap = (folded > math.pi / 4).unsqueeze(-1)
    ctr = torch.where(swap, b[:, [0, 1, 3, 2]], b[:, :4])
    out = torch.zeros((b.shape[0], 5), dtype=b.dtype, device=b.device)
    out[:, :2] = ctr[:, :2] - ctr[:, 2:4] / 2
    out[:, 2:4] = ctr[:, :2] + ctr[:, 2:4] / 2
    return out


def boxes_iou_bev_gpu(boxes_a, boxes_b, box_mode="wlh", metric="rotate_iou", rect=False):
    """(M,7),(N,7) [x,y,z,w,l,h,ry] -> BEV IoU (M,N)  (iou3d_utils.py:32-52). metric 'rotate_iou': rotated rectangles;
    'nearest_iou': their nearest axis-aligned boxes through the same kernel (yaw 0)."""
    if metric == "rotate_iou":
        a_bev, b_bev = boxes3d_to_bev_torch(boxes_a, box_mode, rect), boxes3d_to_bev_torch(boxes_b, box_mode, rect)
    elif metric == "nearest_iou":
        a_bev, b_bev = rbbox2d_to_near_bbox_torch(boxes_a, box_mode, rect), rbbox2d_to_near_bbox_torch(boxes_b, box_mode, rect)
    else:
        raise NotImplementedError
    ans = torch.cuda.FloatTensor(torch.Size((boxes_a.shape[0], boxes_b.shape[0]))).zero_()
    iou3d_cuda.boxes_iou_bev_gpu(a_bev.contiguous(), b_bev.contiguous(), ans)
    return ans


def boxes_iou3d_gpu(boxes_a, boxes_b, box_mode="wlh", rect=False, need_bev=False):
    """(N,7),(M,7) [x,y,z,w,l,h,r] -> 3-D IoU (N,M)  (iou3d_utils.py:143-194)."""
    a_bev, b_bev = boxes3d_to_bev_torch(boxes_a, box_mode, rect), boxes3d_to_bev_torch(boxes_b, box_mode, rect)
    overlaps_bev = torch.cuda.FloatTensor(torch.Size((boxes_a.shape[0], boxes_b.shape[0]))).zero_()
    iou3d_cuda.boxes_overlap_bev_gpu(a_bev.contiguous(), b_bev.contiguous(), overlaps_bev)
    a_hmin, a_hmax = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1), (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    b_hmin, b_hmax = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1), (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_hmax, b_hmax) - torch.max(a_hmin, b_hmin), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    iou3d = overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)
    if need_bev:
        area_a, area_b = (boxes_a[:, 3] * boxes_a[:, 4]).view(-1, 1), (boxes_b[:, 3] * boxes_b[:, 4]).view(1, -1)
        return iou3d, overlaps_bev / torch.clamp(area_a + area_b - overlaps_bev, min=1e-7)
    return iou3d


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b, box_mode="wlh", rect=False, need_bev=False):
    """aligned pairs (N,7),(N,7) -> (N,1)  (iou3d_utils.py:197-252)."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    a_bev, b_bev = boxes3d_to_bev_torch(boxes_a, box_mode, rect), boxes3d_to_bev_torch(boxes_b, box_mode, rect)
    overlaps_bev = torch.cuda.FloatTensor(torch.Size((boxes_a.shape[0], 1))).zero_()
    iou3d_cuda.boxes_aligned_overlap_bev_gpu(a_bev.contiguous(), b_bev.contiguous(), overlaps_bev)
    a_hmin, a_hmax = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1), (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    b_hmin, b_hmax = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(-1, 1), (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(-1, 1)
    overlaps_h = torch.clamp(torch.min(a_hmax, b_hmax) - torch.max(a_hmin, b_hmin), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(-1, 1)
    iou3d = overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)
    if need_bev:
        area_a, area_b = (boxes_a[:, 3] * boxes_a[:, 4]).view(-1, 1), (boxes_b[:, 3] * boxes_b[:, 4]).view(-1, 1)
        return iou3d, overlaps_bev / torch.clamp(area_a + area_b - overlaps_bev, min=1e-7)
    return iou3d


def nms_gpu(boxes, scores, thresh, box_mode="wlh"):
    boxes = boxes3d_to_bev_torch(boxes, box_mode, rect=True)
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = iou3d_cuda.nms_gpu(boxes, keep, thresh)
    return order[keep[:num_out].cuda()].contiguous()


def nms_3d_gpu(boxes, scores, thresh, box_mode="wlh"):
    boxes = boxes3d_to_bev_3d_torch(boxes, box_mode, rect=False)
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = iou3d_cuda.nms_3d_gpu(boxes, keep, thresh)
    return order[keep[:num_out].cuda()].contiguous()


def nms_normal_gpu(boxes, scores, thresh):
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = iou3d_cuda.nms_normal_gpu(boxes, keep, thresh)
    return order[keep[:num_out].cuda()].contiguous()


# ---- host-tensor twins (iou3d_utils.py:7-29,54-120): CPU tensors in, CPU tensors out, through iou3d_cuda's *_cpu entry points
def boxes_iou_bev_cpu(boxes_a, boxes_b, box_mode="wlh", metric="rotate_iou", rect=False):
    """(N,7),(M,7) host tensors -> BEV IoU (N,M) host tensor  (iou3d_utils.py:7-29)."""
    if metric == "rotate_iou":
        a_bev, b_bev = boxes3d_to_bev_torch(boxes_a, box_mode, rect), boxes3d_to_bev_torch(boxes_b, box_mode, rect)
    elif metric == "nearest_iou":
        a_bev, b_bev = rbbox2d_to_near_bbox_torch(boxes_a, box_mode, rect), rbbox2d_to_near_bbox_torch(boxes_b, box_mode, rect)
    else:
        raise NotImplementedError
    iou_bev = torch.FloatTensor(torch.Size((boxes_a.shape[0], boxes_b.shape[0]))).zero_()
    iou3d_cuda.boxes_iou_bev_cpu(a_bev.contiguous(), b_bev.contiguous(), iou_bev)
    return iou_bev


def boxes_iou3d_cpu_test(boxes_a, boxes_b, box_mode="wlh", rect=False):
    """(iou3d_utils.py:54-70) 3-D IoU through the extension's own boxes_iou3d_cpu on [x1,y1,z1,x2,y2,z2,ry] boxes."""
    a3, b3 = boxes3d_to_bev_3d_torch(boxes_a, box_mode, rect), boxes3d_to_bev_3d_torch(boxes_b, box_mode, rect)
    iou3d = torch.FloatTensor(torch.Size((boxes_a.shape[0], boxes_b.shape[0]))).zero_()
    iou3d_cuda.boxes_iou3d_cpu(a3.contiguous(), b3.contiguous(), iou3d)
    return iou3d


def boxes_iou3d_cpu(boxes_a, boxes_b, box_mode="wlh", rect=False, need_bev=False):
    """(iou3d_utils.py:72-120) host tensors; NOTE the reference's host wrapper takes the height range as [z - h, z]
    (the *_gpu wrappers use z -+ h/2) -- reproduced as is."""
    w_index, l_index, h_index = box_mode.index("w") + 3, box_mode.index("l") + 3, box_mode.index("h") + 3
    a_bev, b_bev = boxes3d_to_bev_torch(boxes_a, box_mode, rect), boxes3d_to_bev_torch(boxes_b, box_mode, rect)
    overlaps_bev = torch.FloatTensor(torch.Size((boxes_a.shape[0], boxes_b.shape[0]))).zero_()
    iou3d_cuda.boxes_overlap_bev_cpu(a_bev.contiguous(), b_bev.contiguous(), overlaps_bev)
    area_a = (boxes_a[:, w_index] * boxes_a[:, l_index]).view(-1, 1)
    area_b = (boxes_b[:, w_index] * boxes_b[:, l_index]).view(1, -1)
    iou_bev = overlaps_bev / torch.clamp(area_a + area_b - overlaps_bev, min=1e-7)
    up = 1 if rect else 2
    a_hmin, a_hmax = (boxes_a[:, up] - boxes_a[:, h_index]).view(-1, 1), boxes_a[:, up].view(-1, 1)
    b_hmin, b_hmax = (boxes_b[:, up] - boxes_b[:, h_index]).view(1, -1), boxes_b[:, up].view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_hmax, b_hmax) - torch.max(a_hmin, b_hmin), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, h_index] * boxes_a[:, w_index] * boxes_a[:, l_index]).view(-1, 1)
    vol_b = (boxes_b[:, h_index] * boxes_b[:, w_index] * boxes_b[:, l_index]).view(1, -1)
    iou3d = overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)
    return (iou3d, iou_bev) if need_bev else iou3d
