"""Top-level module `iou3d_cuda`: the torch-extension API of det3d/core/iou3d/src/iou3d.cpp:270-281 (all ten exports), served by
libsessd_hip.so. Same names, argument order, pre-allocated outputs and return values (1 / number kept); inputs
must be contiguous device tensors (TypeError / ValueError instead of the reference's TORCH_CHECK). `keep` is a
CPU LongTensor as in the reference (iou3d.cpp:117-164) -- the only host copy is the kept-index list itself."""
import torch

from sessd_hip import ops


def _check(*ts):
    for t in ts:
        if not t.is_cuda:
            raise ValueError("must be a CUDA tensor ")
        if not t.is_contiguous():
            raise ValueError("must be contiguous ")


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _check(boxes_a, boxes_b, ans_overlap)
    ops.boxes_pairwise(0, boxes_a, boxes_b, ans_overlap)
    return 1


def boxes_aligned_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _check(boxes_a, boxes_b, ans_overlap)
    ops.boxes_aligned_overlap_bev(boxes_a, boxes_b, ans_overlap.view(-1))
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    _check(boxes_a, boxes_b, ans_iou)
    ops.boxes_pairwise(1, boxes_a, boxes_b, ans_iou)
    return 1


def boxes_iou3d_gpu(boxes_a, boxes_b, ans_iou):
    _check(boxes_a, boxes_b, ans_iou)
    ops.boxes_pairwise(2, boxes_a, boxes_b, ans_iou)
    return 1


def _host_twin(mode, boxes_a, boxes_b, out):
    """The *_cpu entry points (iou3d.cpp:275-277 -> iou3d_cpu.cpp:270-336) take HOST tensors and fill a host output. There is no
    CPU implementation in this library: the boxes go to the device, the same kernels run, the result is copied back."""
    for t in (boxes_a, boxes_b, out):
        if t.is_cuda:
            raise ValueError("the _cpu entry points take host tensors")
        if not t.is_contiguous():
            raise ValueError("must be contiguous ")
    dev = torch.device("cuda", torch.cuda.current_device())
    res = ops.boxes_pairwise(mode, boxes_a.float().to(dev), boxes_b.float().to(dev))
    out.copy_(res.cpu().view_as(out))
    return 1


def boxes_overlap_bev_cpu(boxes_a, boxes_b, ans_overlap):
    return _host_twin(0, boxes_a, boxes_b, ans_overlap)


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    return _host_twin(1, boxes_a, boxes_b, ans_iou)


def boxes_iou3d_cpu(boxes_a, boxes_b, ans_iou):
    """(N,7) [x1,y1,z1,x2,y2,z2,ry]; like the reference's host code, disjoint z ranges give overlap * 1e-8 instead of 0."""
    return _host_twin(3, boxes_a, boxes_b, ans_iou)


def _nms(mode, boxes, keep, thresh):
    _check(boxes)
    if not keep.is_contiguous():
        raise ValueError("must be contiguous ")
    k, num = ops.nms_sorted(mode, boxes, thresh)
    n = int(num.item())
    keep[:n] = k[:n].cpu()
    return n


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(0, boxes, keep, nms_overlap_thresh)


def nms_3d_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(1, boxes, keep, nms_overlap_thresh)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(2, boxes, keep, nms_overlap_thresh)
