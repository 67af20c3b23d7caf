"""mirrors det3d/core/iou3d/iou3d_utils.py (GPU functions :32-52,143-306) and utils.py:74-126 converters."""
import torch

import iou3d_cuda


def boxes3d_to_bev_torch(boxes3d, box_mode="wlh", rect=False):
    """[x,y,z,w,l,h,r] -> [x1,y1,x2,y2,ry] (utils.py:74-101)."""
    boxes_bev = boxes3d.new(torch.Size((boxes3d.shape[0], 5)))
    if box_mode == "wlh":
        cu, cv = boxes3d[:, 0], boxes3d[:, 1]
        half_w, half_l = boxes3d[:, 3] / 2, boxes3d[:, 4] / 2
    elif box_mode == "hwl":
        cu, cv = boxes3d[:, 0], (boxes3d[:, 2] if rect else boxes3d[:, 1])
        half_w, half_l = boxes3d[:, 4] / 2, boxes3d[:, 5] / 2
    else:
        raise NotImplementedError
    boxes_bev[:, 0], boxes_bev[:, 1] = cu - half_w, cv - half_l
    boxes_bev[:, 2], boxes_bev[:, 3] = cu + half_w, cv + half_l
    boxes_bev[:, 4] = boxes3d[:, 6]
    return boxes_bev


def boxes3d_to_bev_3d_torch(boxes3d, box_mode="wlh", rect=False):
    """[x,y,z,w,l,h,r] -> [x1,y1,z1,x2,y2,z2,ry] with z +- h/2 (utils.py:104-126)."""
    assert box_mode == "wlh"
    out = boxes3d.new(torch.Size((boxes3d.shape[0], 7)))
    out[:, 0], out[:, 1] = boxes3d[:, 0] - boxes3d[:, 3] / 2, boxes3d[:, 1] - boxes3d[:, 4] / 2
    out[:, 2] = boxes3d[:, 2] - boxes3d[:, 5] / 2
    out[:, 3], out[:, 4] = boxes3d[:, 0] + boxes3d[:, 3] / 2, boxes3d[:, 1] + boxes3d[:, 4] / 2
    out[:, 5] = boxes3d[:, 2] + boxes3d[:, 5] / 2
    out[:, 6] = boxes3d[:, 6]
    return out


def boxes_iou_bev_gpu(boxes_a, boxes_b):
    """(N,5),(M,5) [x1,y1,x2,y2,ry] -> (N,M)  (iou3d_utils.py:32-52)."""
    ans = torch.cuda.FloatTensor(torch.Size((boxes_a.shape[0], boxes_b.shape[0]))).zero_()
    iou3d_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans)
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
