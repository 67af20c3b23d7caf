"""DI-NMS oracle (oracle/di_nms.c, restating det3d/ops/nms/nms_cpu.h:173-384) -- known answers, and the golden of the reference's
numpy wrapper (tests/golden/di_nms_ref.npz, made by make_golden_di_nms.py) reproduced with THIS repository's numpy helpers
(det3d mirror box_np_ops: footprint corners, stand-up boxes, iou_jit) feeding the same core."""
import os
import sys

import numpy as np

from oracle import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _square(cx, cy, s=2.0):
    h = s / 2
    return np.array([[cx - h, cy - h], [cx - h, cy + h], [cx + h, cy + h], [cx + h, cy - h]], np.float32)


def _standup_iou(corners):
    lo, hi = corners.min(1), corners.max(1)
    n = len(corners)
    out = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in range(n):
            w = min(hi[i, 0], hi[j, 0]) - max(lo[i, 0], lo[j, 0])
            h = min(hi[i, 1], hi[j, 1]) - max(lo[i, 1], lo[j, 1])
            if w > 0 and h > 0:
                a = (hi[i] - lo[i]).prod() + (hi[j] - lo[j]).prod() - w * h
                out[i, j] = w * h / a
    return out


def test_known_answer_two_clusters():
    # cluster A: three 2x2 squares at x = 0, 0.5, 1.0 (IoU 0.6 / 1/3 with the first); cluster B: one lonely square far away
    xs = [0.0, 0.5, 1.0, 20.0]
    corners = np.stack([_square(x, 0.0) for x in xs])
    boxes = np.zeros((4, 7), np.float32)
    boxes[:, 0] = xs
    boxes[:, 3:6] = [2, 2, 1.5]
    scores = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    ioup = np.array([1.0, 0.5, 0.5, 1.0], np.float32)
    labels, dirs = np.zeros(4, np.int32), np.array([1, 0, 0, 1], np.int32)
    res = capi.di_nms_core(boxes, corners, _standup_iou(corners), 0.5, scores, ioup, labels, dirs, np.zeros((1, 1)), 1.2,
                           (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, 0)
    b, s, l, d, keep = res
    # pass 1: A = box 0: overlaps 1, 0.6, 1/3 -> cnt = 1 + 0.3 + 1/6 = 1.4667 > 1.2: kept. All three exceed 0.3, weights
    # exp(-(1-ov)^2 / 0.0009) * iou_pred are ~ (1, 0, 0): the average is box 0 itself; score = max normalised score * max = 0.9
    assert keep[0] == 0 and abs(s[0] - 0.9) < 1e-6 and np.allclose(b[0], boxes[0], atol=1e-5) and d[0] == 1
    # boxes 1 and 2 were suppressed by pass 1 (overlap >= 0.3); box 3 alone: cnt = 1 <= 1.2 -> dropped
    assert keep == [0]
    # lowering the count threshold keeps the lonely box as well
    res = capi.di_nms_core(boxes, corners, _standup_iou(corners), 0.5, scores, ioup, labels, dirs, np.zeros((1, 1)), 0.9,
                           (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, 0)
    assert res[4] == [0, 3] and abs(res[1][1] - 0.6) < 1e-6
    # a failed pass gives its suppressed boxes back: with cnt_thresh 1.5 box 0 fails (1.4667), boxes 1 and 2 return and are tried
    # in turn (box 1: 0.6 * 1 + 0.5 + 0.6 * 0.5 = 1.4; box 2: 1/3 + 0.3 + 0.5 = 1.1333): nothing is kept
    res = capi.di_nms_core(boxes, corners, _standup_iou(corners), 0.5, scores, ioup, labels, dirs, np.zeros((1, 1)), 1.5,
                           (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, 0)
    assert res[4] == []
    # a different label in the cluster neither counts nor is averaged, but is still suppressed
    lab2 = np.array([0, 1, 0, 0], np.int32)
    res = capi.di_nms_core(boxes, corners, _standup_iou(corners), 0.5, scores, ioup, lab2, dirs, np.zeros((1, 1)), 1.1,
                           (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, 0)
    assert res[4] == [0]          # cnt = 1 + 1/6 = 1.1667 > 1.1; box 1 (other label) was suppressed by the pass all the same


def test_weighted_average_uses_sigma_of_the_distance_band():
    # far from the origin (40..60 m band, sigma^2 = 0.1) two overlapping boxes are really averaged
    corners = np.stack([_square(45.0, 0.0), _square(45.4, 0.0)])
    boxes = np.zeros((2, 7), np.float32)
    boxes[:, 0] = [45.0, 45.4]
    boxes[:, 3:6] = [2, 2, 1.5]
    res = capi.di_nms_core(boxes, corners, _standup_iou(corners), 0.5, np.array([0.9, 0.5], np.float32), np.array([1.0, 1.0], np.float32),
                           np.zeros(2, np.int32), np.zeros(2, np.int32), np.zeros((1, 1)), 1.0, (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, 0)
    ov = (2 - 0.4) * 2 / (8 - (2 - 0.4) * 2)
    w = np.exp(-(1 - ov) ** 2 / 0.1)
    assert res[4] == [0] and abs(res[0][0][0] - (45.0 + w * 45.4) / (1 + w)) < 1e-4
    # beyond the last band no weight is defined: the reference divides 0 by 0
    boxes[:, 0] += 30
    corners[:, :, 0] += 30
    res = capi.di_nms_core(boxes, corners, _standup_iou(corners), 0.5, np.array([0.9, 0.5], np.float32), np.array([1.0, 1.0], np.float32),
                           np.zeros(2, np.int32), np.zeros(2, np.int32), np.zeros((1, 1)), 1.0, (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, 0)
    assert res[4] == [0] and np.isnan(res[0][0][0])


def test_numpy_wrapper_golden_with_the_mirror_helpers(golden_dir):
    """rotate_weighted_nms_cc (nms_cpu.py:52-93): the reference's helpers + the oracle core gave the golden; the mirror's numpy
    helpers + the same core must reproduce it."""
    from make_golden_di_nms import make_case
    from det3d.core.bbox import box_np_ops
    g = np.load(os.path.join(golden_dir, "di_nms_ref.npz"))
    box, anchors, scores, iou_preds, labels, dirs = make_case(9, n=180)
    dets = np.concatenate([box[:, [0, 1, 3, 4, 6]], scores[:, None]], 1).astype(np.float32)
    corners = box_np_ops.center_to_corner_box2d(dets[:, :2], dets[:, 2:4], dets[:, 4])
    standup = box_np_ops.corner_to_standup_nd(corners)
    sio = box_np_ops.iou_jit(standup, standup, eps=0.0)
    for tag, an, cc in (("cc0", np.zeros((1, 1)), 0), ("cc1", anchors, 1)):
        r = capi.di_nms_core(box, corners, sio, 0.5, dets[:, 5], iou_preds, labels.astype(np.int32), dirs.astype(np.int32), an, 2.6,
                             (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, cc)
        assert r[4] == g[tag + "_keep"].tolist() and len(r[4]) >= 10
        assert np.allclose(np.array(r[0]), g[tag + "_boxes"], atol=1e-5, equal_nan=True) and np.allclose(np.array(r[1]), g[tag + "_scores"], atol=1e-6)
        assert r[2] == g[tag + "_labels"].tolist() and r[3] == g[tag + "_dirs"].tolist()
