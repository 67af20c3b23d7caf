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


def test_iou3d_cuda_host_twins_vs_compiled_reference(dev, golden_dir):
    """boxes_overlap_bev_cpu / boxes_iou_bev_cpu / boxes_iou3d_cpu (iou3d.cpp:275-277): host tensors in, host tensor filled, against
    the outputs of the COMPILED reference iou3d_cpu.cpp stored in iou3d_ref.npz (and live against oracle/_ref where it exists)."""
    import os
    import iou3d_cuda
    g = np.load(os.path.join(golden_dir, "iou3d_ref.npz"))
    a5, b5, a7, b7 = (torch.from_numpy(g[k]) for k in ("a5", "b5", "a7", "b7"))
    n, m = a5.shape[0], b5.shape[0]
    out = torch.FloatTensor(torch.Size((n, m))).zero_()
    assert iou3d_cuda.boxes_overlap_bev_cpu(a5, b5, out) == 1 and not out.is_cuda
    assert np.abs(out.numpy() - g["overlap"]).max() < 2e-4
    out.zero_()
    assert iou3d_cuda.boxes_iou_bev_cpu(a5, b5, out) == 1
    assert np.abs(out.numpy() - g["iou_bev"]).max() < 2e-5
    out.zero_()
    assert iou3d_cuda.boxes_iou3d_cpu(a7, b7, out) == 1
    assert np.abs(out.numpy() - g["iou3d_cpu"]).max() < 2e-5
    # the host twin has no early zero for disjoint height ranges (iou3d_cpu.cpp:322-333): overlap * 1e-8 / union
    lo = np.array([[0, 0, 0, 2, 2, 1, 0.1]], np.float32)
    hi = np.array([[0.5, 0.5, 5, 2.5, 2.5, 6, 0.2]], np.float32)
    o1 = torch.zeros(1, 1)
    iou3d_cuda.boxes_iou3d_cpu(torch.from_numpy(lo), torch.from_numpy(hi), o1)
    o2 = torch.cuda.FloatTensor(1, 1).zero_()
    iou3d_cuda.boxes_iou3d_gpu(torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev), o2)
    assert float(o2) == 0.0 and 0.0 < float(o1) < 1e-7
    assert abs(float(o1) - float(oracle.boxes_iou3d(lo, hi, gpu_variant=False)[0, 0])) < 1e-12
    if oracle.ref_lib() is not None:
        assert np.abs(out.numpy() - oracle.ref_boxes_iou3d(g["a7"], g["b7"])).max() < 2e-5
    with pytest.raises(ValueError):
        iou3d_cuda.boxes_iou_bev_cpu(a5.to(dev), b5, out)  # host entry point, device tensor


def test_iou3d_utils_host_wrappers(dev):
    """iou3d_utils.boxes_iou_bev_cpu / boxes_iou3d_cpu / boxes_iou3d_cpu_test (iou3d_utils.py:7-29,54-120), host tensors: the
    host 3-D wrapper measures the height range as [z - h, z], the extension's own 3-D function z -+ h/2."""
    from det3d.core.iou3d import iou3d_utils as U
    a7, b7 = synth.clustered_boxes7(50, seed=8), synth.clustered_boxes7(40, seed=9)
    A, B = torch.from_numpy(a7), torch.from_numpy(b7)
    a5, b5 = synth.boxes7_to_bev5(a7), synth.boxes7_to_bev5(b7)
    bev = U.boxes_iou_bev_cpu(A, B)
    assert not bev.is_cuda and np.abs(bev.numpy() - oracle.boxes_iou_bev(a5, b5)).max() < 2e-5
    ov = oracle.boxes_overlap_bev(a5, b5)
    hmin = np.maximum((a7[:, 2] - a7[:, 5])[:, None], (b7[:, 2] - b7[:, 5])[None, :])
    hmax = np.minimum(a7[:, 2][:, None], b7[:, 2][None, :])
    o3 = ov * np.clip(hmax - hmin, 0, None)
    want = o3 / np.clip((a7[:, 3] * a7[:, 4] * a7[:, 5])[:, None] + (b7[:, 3] * b7[:, 4] * b7[:, 5])[None, :] - o3, 1e-7, None)
    got, got_bev = U.boxes_iou3d_cpu(A, B, need_bev=True)
    assert np.abs(got.numpy() - want).max() < 5e-5 and np.abs(got_bev.numpy() - oracle.boxes_iou_bev(a5, b5)).max() < 5e-5
    t3 = U.boxes_iou3d_cpu_test(A, B).numpy()
    assert np.abs(t3 - oracle.boxes_iou3d(synth.boxes7_to_bev7(a7), synth.boxes7_to_bev7(b7), gpu_variant=False)).max() < 2e-5


def test_spconv_utils_rbbox_iou_and_box_np_ops_riou_cc(dev):
    """spconv.utils.rbbox_iou / rbbox_intersection (box_np_ops.py:9) on the device quad clipper, called the way riou_cc /
    rinter_cc call them (box_np_ops.py:20-50): corners + stand-up IoU prefilter from the numpy helpers."""
    import spconv
    from oracle import capi
    d = synth.clustered_boxes7(60, seed=13)[:, [0, 1, 3, 4, 6]].astype(np.float32)
    q = d[:45].copy()  # the same boxes, shifted and turned a little: plenty of partial overlaps
    rq = np.random.RandomState(14)
    q[:, :2] += rq.uniform(-0.8, 0.8, (45, 2)).astype(np.float32)
    q[:, 4] += rq.uniform(-0.4, 0.4, 45).astype(np.float32)
    ca, cb = capi.box2d_corners(d), capi.box2d_corners(q)
    sa = np.concatenate([ca.min(1), ca.max(1)], 1)
    sb = np.concatenate([cb.min(1), cb.max(1)], 1)
    iw = np.minimum(sa[:, None, 2], sb[None, :, 2]) - np.maximum(sa[:, None, 0], sb[None, :, 0])
    ih = np.minimum(sa[:, None, 3], sb[None, :, 3]) - np.maximum(sa[:, None, 1], sb[None, :, 1])
    inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
    standup = (inter / (((sa[:, 2] - sa[:, 0]) * (sa[:, 3] - sa[:, 1]))[:, None] + ((sb[:, 2] - sb[:, 0]) * (sb[:, 3] - sb[:, 1]))[None, :] - inter)).astype(np.float32)
    got = spconv.utils.rbbox_iou(ca, cb, standup, 0.0)
    gin = spconv.utils.rbbox_intersection(ca, cb, standup, 0.0)
    want = np.zeros((60, 45), np.float32)
    for i in range(60):
        for j in range(45):
            if standup[i, j] > 0:
                want[i, j] = capi.quad_iou(ca[i], cb[j])
    assert got.shape == (60, 45) and np.abs(got - want).max() < 2e-5 and (want > 0).sum() > 20
    area = lambda c: 0.5 * np.abs(np.sum(c[:, 0] * np.roll(c[:, 1], -1) - np.roll(c[:, 0], -1) * c[:, 1]))
    for i, j in zip(*np.nonzero(want > 0.05)):
        u = area(ca[i]) + area(cb[j]) - gin[i, j]
        assert abs(gin[i, j] / u - want[i, j]) < 5e-5
    # a threshold above every stand-up IoU switches everything off
    assert not spconv.utils.rbbox_iou(ca, cb, standup, 2.0).any()
    from det3d.core.bbox import box_np_ops
    if hasattr(box_np_ops, "riou_cc"):
        r = box_np_ops.riou_cc(d, q)
        assert np.abs(r - want).max() < 5e-5


def test_sparse_conv_tensor_dense_is_a_hip_scatter_with_gradient(dev):
    import spconv
    rng = np.random.RandomState(2)
    B, shape, n, C = 2, [3, 10, 12], 150, 16
    lin = rng.choice(B * 3 * 10 * 12, n, replace=False)
    idx = np.stack([lin // 360, (lin // 120) % 3, (lin // 12) % 10, lin % 12], 1).astype(np.int32)
    feat = torch.randn(n, C, device=dev, requires_grad=True)
    x = spconv.SparseConvTensor(feat, torch.from_numpy(idx).to(dev), shape, B)
    d = x.dense()
    assert tuple(d.shape) == (B, C, 3, 10, 12)
    ref = torch.zeros(B, 3, 10, 12, C, device=dev)
    i = torch.from_numpy(idx).long().to(dev)
    ref[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = feat.detach()
    assert torch.equal(d.detach(), ref.permute(0, 4, 1, 2, 3))
    w = torch.randn_like(d)
    (d * w).sum().backward()
    assert torch.equal(feat.grad, w.permute(0, 2, 3, 4, 1)[i[:, 0], i[:, 1], i[:, 2], i[:, 3]])
    assert torch.equal(x.dense(channels_first=False), ref)
