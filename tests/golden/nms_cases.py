"""Shared by make_golden_nms_cpu.py and tests/test_nms_cpu_ref_cpu.py: seeded clustered boxes for the CPU NMS cores and the
stand-up (axis-aligned hull) IoU matrix both receive as an argument (box_np_ops.iou_jit semantics with eps = 0)."""
import numpy as np

SIGMA_DIST_INTERVAL = np.array([0, 20, 40, 60], np.float32)   # config.py test_cfg.nms_sigma_dist_interval
SIGMA_SQUARE = np.array([0.0009, 0.009, 0.1, 1], np.float32)  # config.py test_cfg.nms_sigma_square
THRESHOLDS = (0.01, 0.3, 0.5)


def make_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 64))
    cen = rng.uniform(0, 60, (max(2, n // 8), 2))
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = cen[rng.integers(0, len(cen), n)] + rng.normal(0, 0.5, (n, 2))
    b[:, 2] = rng.uniform(-2, 0, n)
    b[:, 3], b[:, 4], b[:, 5] = rng.uniform(1.4, 2.0, n), rng.uniform(3.2, 4.6, n), rng.uniform(1.3, 1.8, n)
    b[:, 6] = rng.uniform(-3.14, 3.14, n)
    anchors = b.copy()
    anchors[:, :2] += rng.normal(0, 0.3, (n, 2)).astype(np.float32)
    return dict(boxes=b, scores=rng.uniform(0.3, 1.0, n).astype(np.float32), iou_preds=rng.uniform(0.2, 1.0, n).astype(np.float32),
                labels=(rng.integers(0, 2, n) if seed % 3 == 0 else np.zeros(n)).astype(np.int32), dirs=rng.integers(0, 2, n).astype(np.int32),
                anchors=anchors, cnt_thresh=np.float32(2.6 if seed % 4 else 0.8), centerness_c=np.int32(seed % 2))


def standup_iou(corners):
    """corners (n,4,2) -> (n,n) IoU of the axis-aligned hulls."""
    mn, mx = corners.min(1), corners.max(1)
    iw = np.minimum(mx[:, None, 0], mx[None, :, 0]) - np.maximum(mn[:, None, 0], mn[None, :, 0])
    ih = np.minimum(mx[:, None, 1], mx[None, :, 1]) - np.maximum(mn[:, None, 1], mn[None, :, 1])
    inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
    area = (mx[:, 0] - mn[:, 0]) * (mx[:, 1] - mn[:, 1])
    return (inter / (area[:, None] + area[None, :] - inter)).astype(np.float32)


AXIS_CASES = ((300, 0.5, 1.0), (300, 0.1, 0.0), (120, 0.3, 0.0), (1, 0.5, 0.0))  # (boxes, threshold, eps) of nms_cpu.h:24-70


def make_axis_case(n):
    rng = np.random.RandomState(n + 1)
    xy = rng.uniform(0, 60, (n, 2)).astype(np.float32)
    wh = rng.uniform(2, 12, (n, 2)).astype(np.float32)
    dets = np.concatenate([xy, xy + wh, rng.permutation(n).astype(np.float32)[:, None] / max(n, 1)], 1).astype(np.float32)
    return dets, dets[:, 4].argsort()[::-1].astype(np.int32)
