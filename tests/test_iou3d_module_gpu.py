"""The module-level surfaces of SURVEY 8(b): top-level `iou3d_cuda` (iou3d.cpp:270-281 calling convention: caller-allocated,
zero-filled outputs; CPU LongTensor `keep`; return 1 / number kept) and the `det3d.core.iou3d.iou3d_utils` wrappers
(iou3d_utils.py:32-52,143-306) on (N,7) [x,y,z,w,l,h,ry] boxes, against the CPU oracle (oracle/iou3d.c, bit-equal to the
compiled reference iou3d_cpu.cpp) composed the way the wrappers compose it."""
import numpy as np
import pytest
import torch

import oracle
from sessd_hip import synth

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def test_iou3d_cuda_calling_convention(dev):
    import iou3d_cuda
    a7, b7 = synth.clustered_boxes7(90, seed=1), synth.clustered_boxes7(70, seed=2)
    a5, b5 = synth.boxes7_to_bev5(a7), synth.boxes7_to_bev5(b7)
    out = torch.cuda.FloatTensor(torch.Size((90, 70))).zero_()
    assert iou3d_cuda.boxes_overlap_bev_gpu(_t(a5, dev), _t(b5, dev), out) == 1
    assert np.abs(out.cpu().numpy() - oracle.boxes_overlap_bev(a5, b5)).max() < 2e-4
    out.zero_()
    assert iou3d_cuda.boxes_iou_bev_gpu(_t(a5, dev), _t(b5, dev), out) == 1
    assert np.abs(out.cpu().numpy() - oracle.boxes_iou_bev(a5, b5)).max() < 2e-5
    out.zero_()
    assert iou3d_cuda.boxes_iou3d_gpu(_t(synth.boxes7_to_bev7(a7), dev), _t(synth.boxes7_to_bev7(b7), dev), out) == 1
    assert np.abs(out.cpu().numpy() - oracle.boxes_iou3d(synth.boxes7_to_bev7(a7), synth.boxes7_to_bev7(b7))).max() < 2e-5
    al = torch.cuda.FloatTensor(torch.Size((70, 1))).zero_()
    assert iou3d_cuda.boxes_aligned_overlap_bev_gpu(_t(a5[:70], dev), _t(b5, dev), al) == 1
    assert np.abs(al.cpu().numpy().reshape(-1) - oracle.boxes_aligned_overlap_bev(a5[:70], b5)).max() < 2e-4
    # NMS: boxes sorted by the caller, CPU keep tensor, count returned
    srt = synth.boxes7_to_bev5(synth.clustered_boxes7(300, seed=5))
    keep = torch.LongTensor(300)
    n = iou3d_cuda.nms_gpu(_t(srt, dev), keep, 0.1)
    want = oracle.nms_sorted(srt, 0.1, 0)
    assert not keep.is_cuda and n == len(want) and np.array_equal(keep[:n].numpy(), want)
    with pytest.raises(ValueError):
        iou3d_cuda.boxes_iou_bev_gpu(torch.zeros(3, 5), _t(b5, dev), out)  # CHECK_INPUT: must be a device tensor


def test_iou3d_utils_wrappers(dev):
    from det3d.core.iou3d import iou3d_utils as U
    a7, b7 = synth.clustered_boxes7(80, seed=3), synth.clustered_boxes7(60, seed=4)
    A, B = _t(a7, dev), _t(b7, dev)
    a5, b5 = synth.boxes7_to_bev5(a7), synth.boxes7_to_bev5(b7)
    bev = U.boxes_iou_bev_gpu(A, B).cpu().numpy()                       # (N,7) in, conversion inside (iou3d_utils.py:32-52)
    assert bev.shape == (80, 60) and np.abs(bev - oracle.boxes_iou_bev(a5, b5)).max() < 2e-5
    near = U.boxes_iou_bev_gpu(A, B, metric="nearest_iou").cpu().numpy()
    na, nb = U.rbbox2d_to_near_bbox_torch(A).cpu().numpy(), U.rbbox2d_to_near_bbox_torch(B).cpu().numpy()
    assert np.abs(near - oracle.boxes_iou_bev(na, nb)).max() < 2e-5 and np.all(na[:, 4] == 0)
    # 3-D IoU = BEV overlap x height overlap / union volume, z from centre +- h/2 (iou3d_utils.py:152-194)
    ov = oracle.boxes_overlap_bev(a5, b5)
    hmin = np.maximum((a7[:, 2] - a7[:, 5] / 2)[:, None], (b7[:, 2] - b7[:, 5] / 2)[None, :])
    hmax = np.minimum((a7[:, 2] + a7[:, 5] / 2)[:, None], (b7[:, 2] + b7[:, 5] / 2)[None, :])
    o3 = ov * np.clip(hmax - hmin, 0, None)
    want3 = o3 / np.clip((a7[:, 3] * a7[:, 4] * a7[:, 5])[:, None] + (b7[:, 3] * b7[:, 4] * b7[:, 5])[None, :] - o3, 1e-7, None)
    got3, got_bev = U.boxes_iou3d_gpu(A, B, need_bev=True)
    assert np.abs(got3.cpu().numpy() - want3).max() < 5e-5 and np.abs(got_bev.cpu().numpy() - oracle.boxes_iou_bev(a5, b5)).max() < 5e-5
    al = U.boxes_aligned_iou3d_gpu(A[:60], B).cpu().numpy()
    assert al.shape == (60, 1) and np.abs(al.reshape(-1) - np.diag(want3[:60])).max() < 5e-5
    # NMS wrappers: indices into the UNSORTED input, on the device
    scores = torch.from_numpy(np.random.RandomState(0).permutation(80).astype(np.float32)).to(dev)
    order = torch.argsort(scores, descending=True).cpu().numpy()
    k = U.nms_gpu(A, scores, 0.1)
    # nms_gpu converts with rect=True (iou3d_utils.py:263): footprint (x, z) with half extents (l/2, w/2) (utils.py:91-94)
    bev_rect = np.stack([a7[:, 0] - a7[:, 4] / 2, a7[:, 2] - a7[:, 3] / 2, a7[:, 0] + a7[:, 4] / 2, a7[:, 2] + a7[:, 3] / 2,
                         a7[:, 6]], 1).astype(np.float32)
    assert k.is_cuda and np.array_equal(k.cpu().numpy(), order[oracle.nms_sorted(bev_rect[order], 0.1, 0)])
    k3 = U.nms_3d_gpu(A, scores, 0.1)
    assert np.array_equal(k3.cpu().numpy(), order[oracle.nms_sorted(synth.boxes7_to_bev7(a7)[order], 0.1, 1)])
    kn = U.nms_normal_gpu(_t(a5, dev), scores, 0.1)
    assert np.array_equal(kn.cpu().numpy(), order[oracle.nms_sorted(a5[order], 0.1, 2)])
