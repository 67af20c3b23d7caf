"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the anchor target assignment of the SE-SSD training pipeline.

Restates, vectorised in numpy float32,
    det3d/core/anchor/target_ops_v3.py:11-137       create_target_np
    det3d/core/bbox/region_similarity.py:85-98      NearestIouSimilarity._compare
    det3d/core/bbox/box_np_ops.py:354-366,619-620   rbbox2d_to_near_bbox, limit_period ; :1008-1046 iou_jit(eps=0) ; :52-110 second_box_encode
as det3d/core/anchor/target_assigner.py:68-136 wires them (no anchor pruning, no positive-fraction subsampling).
Pinned by tests/golden/assign_ref.npz = the reference functions run from source (tests/golden/make_golden_assign.py)."""
import numpy as np

F = np.float32


def near_bbox(rb):
    """[x, y, w, l, r] -> nearest axis-aligned [x1, y1, x2, y2]: sizes swap when |limit_period(r, 0.5, pi)| > pi/4."""
    rb = np.asarray(rb, F)
    r = rb[:, 4]
    folded = np.abs(r - np.floor(r / F(np.pi) + F(0.5)) * F(np.pi))
    swap = folded > F(np.pi / 4)
    w = np.where(swap, rb[:, 3], rb[:, 2])
    l = np.where(swap, rb[:, 2], rb[:, 3])
    return np.stack([rb[:, 0] - w / F(2), rb[:, 1] - l / F(2), rb[:, 0] + w / F(2), rb[:, 1] + l / F(2)], 1).astype(F)


def nearest_iou(anchors7, gt7):
    a, g = near_bbox(anchors7[:, [0, 1, 3, 4, 6]]), near_bbox(gt7[:, [0, 1, 3, 4, 6]])
    iw = np.minimum(a[:, None, 2], g[None, :, 2]) - np.maximum(a[:, None, 0], g[None, :, 0])
    ih = np.minimum(a[:, None, 3], g[None, :, 3]) - np.maximum(a[:, None, 1], g[None, :, 1])
    ok = (iw > 0) & (ih > 0)
    inter = (iw * ih).astype(F)
    area_a = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])).astype(F)
    area_g = ((g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])).astype(F)
    ua = (area_a[:, None] + area_g[None, :] - inter).astype(F)
    return np.where(ok, inter / np.where(ok, ua, F(1)), F(0)).astype(F)


def box_encode(boxes, anchors):
    xa, ya, za, wa, la, ha, ra = [anchors[:, i] for i in range(7)]
    xg, yg, zg, wg, lg, hg, rg = [boxes[:, i] for i in range(7)]
    diag = np.sqrt(la ** 2 + wa ** 2)
    return np.stack([(xg - xa) / diag, (yg - ya) / diag, (zg - za) / ha, np.log(wg / wa), np.log(lg / la), np.log(hg / ha),
                     rg - ra], 1).astype(F)


def assign(anchors, gt_boxes, gt_classes=None, matched=0.6, unmatched=0.45):
    """Returns dict(labels (N,) int32 in {-1,0,class}, bbox_targets (N,7) f32, bbox_outside_weights (N,) f32,
    positive_gt_id (P,) int32 in anchor order)."""
    anchors, gt_boxes = np.asarray(anchors, F), np.asarray(gt_boxes, F).reshape(-1, 7)
    n, m = anchors.shape[0], gt_boxes.shape[0]
    gt_classes = np.ones(m, np.int32) if gt_classes is None else np.asarray(gt_classes, np.int32)
    labels = np.full(n, -1, np.int32)
    gt_ids = np.full(n, -1, np.int32)
    targets = np.zeros((n, 7), F)
    if m == 0:
        labels[:] = 0
        return dict(labels=labels, bbox_targets=targets, bbox_outside_weights=np.zeros(n, F), positive_gt_id=np.zeros(0, np.int32))
    iou = nearest_iou(anchors, gt_boxes)
    a_arg = iou.argmax(1)
    a_max = iou[np.arange(n), a_arg]
    g_max = iou.max(0).copy()
    g_max[g_max == 0] = -1                         # a box that overlaps no anchor forces nothing
    force = np.nonzero((iou == g_max[None, :]).any(1))[0]
    labels[force] = gt_classes[a_arg[force]]
    gt_ids[force] = a_arg[force]
    pos = a_max >= F(matched)
    labels[pos] = gt_classes[a_arg[pos]]
    gt_ids[pos] = a_arg[pos]
    fg = np.nonzero(labels > 0)[0]                 # taken BEFORE the background pass, as the reference does
    labels[a_max < F(unmatched)] = 0
    labels[force] = gt_classes[a_arg[force]]       # forced positives survive the background pass
    targets[fg] = box_encode(gt_boxes[a_arg[fg]], anchors[fg])
    w = np.zeros(n, F)
    w[labels > 0] = 1.0
    return dict(labels=labels, bbox_targets=targets, bbox_outside_weights=w, positive_gt_id=gt_ids[fg])
