"""HIP iou3d operators vs the compiled-reference golden vectors and the CPU oracle.

Tolerance (float32 geometry): |area| error <= 2e-4 m^2 and |IoU| error <= 2e-5 -- the device
sinf/cosf/atan2f are not glibc's, everything else is the same operation order as the reference.
NMS keep lists must be identical unless a pair sits within 1e-4 of the threshold."""
import os

import numpy as np
import pytest
import torch

import oracle
from sessd_hip import ops, synth

pytestmark = pytest.mark.gpu
ATOL_AREA, ATOL_IOU = 2e-4, 2e-5


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def test_golden_pairwise(golden_dir, dev):
    g = np.load(os.path.join(golden_dir, "iou3d_ref.npz"))
    ov = ops.boxes_pairwise(0, _t(g["a5"], dev), _t(g["b5"], dev)).cpu().numpy()
    assert np.abs(ov - g["overlap"]).max() < ATOL_AREA
    assert ((ov > 0) == (g["overlap"] > 0)).mean() > 0.999
    iou = ops.boxes_pairwise(1, _t(g["a5"], dev), _t(g["b5"], dev)).cpu().numpy()
    assert np.abs(iou - g["iou_bev"]).max() < ATOL_IOU
    lit = ops.boxes_pairwise(1, _t(g["literal"], dev), _t(g["literal"], dev)).cpu().numpy()
    assert np.abs(lit - g["lit_iou"]).max() < ATOL_IOU
    assert abs(lit[0, 1] - 1 / 7) < 1e-5 and abs(lit[0, 2] - 0.70710678) < 1e-5 and lit[0, 3] == 0.0


@pytest.mark.parametrize("n,m", [(1, 1), (17, 33), (128, 128), (1000, 257)])
def test_pairwise_vs_oracle(dev, n, m):
    a7, b7 = synth.clustered_boxes7(n, seed=n), synth.clustered_boxes7(m, seed=m + 1, clusters=max(1, n // 12))
    a5, b5 = synth.boxes7_to_bev5(a7), synth.boxes7_to_bev5(b7)
    assert np.abs(ops.boxes_pairwise(0, _t(a5, dev), _t(b5, dev)).cpu().numpy() - oracle.boxes_overlap_bev(a5, b5)).max() < ATOL_AREA
    assert np.abs(ops.boxes_pairwise(1, _t(a5, dev), _t(b5, dev)).cpu().numpy() - oracle.boxes_iou_bev(a5, b5)).max() < ATOL_IOU
    a3, b3 = synth.boxes7_to_bev7(a7), synth.boxes7_to_bev7(b7)
    got = ops.boxes_pairwise(2, _t(a3, dev), _t(b3, dev)).cpu().numpy()
    assert np.abs(got - oracle.boxes_iou3d(a3, b3, gpu_variant=True)).max() < ATOL_IOU


def test_aligned_and_empty(dev):
    a7, b7 = synth.clustered_boxes7(300, seed=1), synth.clustered_boxes7(300, seed=1)
    b7[:, :2] += 0.3
    a5, b5 = synth.boxes7_to_bev5(a7), synth.boxes7_to_bev5(b7)
    got = ops.boxes_aligned_overlap_bev(_t(a5, dev), _t(b5, dev)).cpu().numpy()
    assert np.abs(got - oracle.boxes_aligned_overlap_bev(a5, b5)).max() < ATOL_AREA
    e = ops.boxes_pairwise(1, _t(a5[:0], dev), _t(b5, dev))
    assert e.shape == (0, 300)


def test_symmetry_and_identity_properties(dev):
    a5 = synth.boxes7_to_bev5(synth.clustered_boxes7(1000, seed=3))
    iou = ops.boxes_pairwise(1, _t(a5, dev), _t(a5, dev)).cpu().numpy()
    assert np.abs(np.diag(iou) - 1).max() < 1e-4
    assert np.abs(iou - iou.T).max() < 1e-4
    assert iou.min() >= 0 and iou.max() <= 1 + 1e-4


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n,thresh", [(1, 0.1), (64, 0.01), (300, 0.1), (1000, 0.7), (1500, 0.3)])
def test_nms_vs_oracle(dev, mode, n, thresh):
    b7 = synth.clustered_boxes7(n, seed=n + mode)
    boxes = synth.boxes7_to_bev7(b7) if mode == 1 else synth.boxes7_to_bev5(b7)
    keep, num = ops.nms_sorted(mode, _t(boxes, dev), thresh)
    k = int(num.item())
    got = keep[:k].cpu().numpy()
    want = oracle.nms_sorted(boxes, thresh, mode)
    if not np.array_equal(got, want):
        # only legitimate when a decisive pair sits on the threshold
        fn = {0: oracle.boxes_iou_bev, 2: None}.get(mode)
        iou = oracle.boxes_iou_bev(boxes, boxes) if mode == 0 else None
        near = None if iou is None else int((np.abs(iou - thresh) < 1e-4).sum())
        pytest.fail("NMS keep differs (near-threshold pairs: %s)" % near)
