"""mirrors det3d/datasets/utils/eval.py: the building blocks of the KITTI average-precision evaluation.
Overlap matrices: 2-D image boxes :282-312, BEV :315-321 and 3-D :324-367 -- the rotated overlaps come from the device
(det3d.ops.nms.nms_gpu.rotate_iou_gpu_eval -> sessd_rotate_iou_eval; box3d_overlap -> sessd_box3d_overlap_eval in one launch; the
reference uses a numba-CUDA kernel + a numba loop). Matching statistics of one frame :144-278: `compute_statistics_jit` below is
the host form of the sequential greedy assignment (the reference's structure, pinned by tests/golden/kitti_eval_ref.npz); the
evaluation itself runs it on the device for all (frame, threshold) pairs at once (det3d.datasets.kitti.eval, sessd_kitti_*)."""
import numpy as np

from det3d.ops.nms.nms_gpu import rotate_iou_gpu_eval


FUSED_BOX3D_OVERLAP = True


def get_split_parts(num, num_part):
    same, rest = num // num_part, num % num_part
    return [same] * num_part + ([rest] if rest else [])


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """(N,4),(K,4) [x1,y1,x2,y2] -> (N,K): intersection over union (-1), over area(box) (0), over area(query) (1), raw (else)."""
    b, q = np.asarray(boxes), np.asarray(query_boxes)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=b.dtype)
    if out.size == 0:
        return out
    iw = np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0])
    ih = np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1])
    ok = (iw > 0) & (ih > 0)
    inter = iw * ih
    area_b = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[:, None]
    area_q = ((q[:, 2] - q[:, 0]) * (q[:, 3] - q[:, 1]))[None, :]
    if criterion == -1:
        ua = area_b + area_q - inter
    elif criterion == 0:
        ua = np.broadcast_to(area_b, inter.shape)
    elif criterion == 1:
        ua = np.broadcast_to(area_q, inter.shape)
    else:
        ua = np.ones_like(inter)
    out[ok] = (inter[ok] / ua[ok]).astype(b.dtype)
    return out


def bev_box_overlap(boxes, qboxes, criterion=-1, stable=False):
    return rotate_iou_gpu_eval(boxes, qboxes, criterion)


def box3d_overlap(boxes, qboxes, criterion=-1, z_axis=1, z_center=1.0, fused=None):
    """(N,7),(K,7) [loc3, dims3, rot] -> 3-D overlap: rotated BEV intersection (criterion 2) x height overlap, normalised by
    union (-1) / volume(box) (0) / volume(query) (1). z_axis = index of the height axis (KITTI camera: 1), z_center = where
    the location sits inside the box height (camera: 1.0 = bottom face at `location`). ONE device launch
    (sessd_box3d_overlap_eval: float32 rotated part, float64 height / volume part, like the reference's two steps);
    fused=False keeps the numpy composition around rotate_iou_gpu_eval's intersections (the reference's two-step structure:
    the cross-check of the fused kernel, and what the host-side tests drive with the oracle's intersections)."""
    import torch
    if fused is None:
        fused = FUSED_BOX3D_OVERLAP and torch.cuda.is_available()
    if fused:
        from sessd_hip import ops
        if boxes.shape[0] == 0 or qboxes.shape[0] == 0:
            return np.zeros((boxes.shape[0], qboxes.shape[0]), dtype=boxes.dtype)
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(dev)
        return ops.box3d_overlap_eval(t(boxes), t(qboxes), criterion, z_axis, z_center).cpu().numpy().astype(boxes.dtype)
    bev = list(range(7))
    bev.pop(z_axis + 3)
    bev.pop(z_axis)
    rinc = rotate_iou_gpu_eval(boxes[:, bev], qboxes[:, bev], 2)
    if rinc.size == 0:
        return rinc
    zb, hb = boxes[:, z_axis][:, None], boxes[:, z_axis + 3][:, None]
    zq, hq = qboxes[:, z_axis][None, :], qboxes[:, z_axis + 3][None, :]
    dz = np.minimum(zb + hb * (1 - z_center), zq + hq * (1 - z_center)) - np.maximum(zb - hb * z_center, zq - hq * z_center)
    vol_b = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
    vol_q = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
    inc = dz * rinc
    if criterion == -1:
        ua = vol_b + vol_q - inc
    elif criterion == 0:
        ua = np.broadcast_to(vol_b, inc.shape)
    elif criterion == 1:
        ua = np.broadcast_to(vol_q, inc.shape)
    else:
        ua = np.ones_like(inc)
    hit = rinc > 0
    out = np.where(hit & (dz > 0), inc / np.where(ua == 0, 1, ua), 0.0)
    return np.where(hit, out, rinc).astype(rinc.dtype)


def _boxes_of(annos, metric, bev_axes):
    if metric == 0:
        return np.concatenate([a["bbox"] for a in annos], 0)
    cols = bev_axes if metric == 1 else [0, 1, 2]
    loc = np.concatenate([a["location"][:, cols] for a in annos], 0)
    dims = np.concatenate([a["dimensions"][:, cols] for a in annos], 0)
    rots = np.concatenate([a["rotation_y"] for a in annos], 0)
    return np.concatenate([loc, dims, rots[..., np.newaxis]], axis=1)


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50, z_axis=1, z_center=1.0):
    """Overlap matrices per frame, computed over `num_parts` groups of frames at a time (one device call per group).
    metric 0: 2-D bbox, 1: BEV, 2: 3-D. Returns (per-frame overlaps, per-group overlaps, boxes per frame of each side)."""
    assert len(gt_annos) == len(dt_annos)
    n_dt = np.stack([len(a["name"]) for a in dt_annos], 0)
    n_gt = np.stack([len(a["name"]) for a in gt_annos], 0)
    parts = [p for p in get_split_parts(len(gt_annos), num_parts) if p != 0]
    bev_axes = [i for i in range(3) if i != z_axis]
    parted, start = [], 0
    for p in parts:
        g, d = gt_annos[start:start + p], dt_annos[start:start + p]
        gb, db = _boxes_of(g, metric, bev_axes), _boxes_of(d, metric, bev_axes)
        if metric == 0:
            ov = image_box_overlap(gb, db)
        elif metric == 1:
            ov = bev_box_overlap(gb, db).astype(np.float64)
        elif metric == 2:
            ov = box3d_overlap(gb, db, z_axis=z_axis, z_center=z_center).astype(np.float64)
        else:
            raise ValueError("unknown metric")
        parted.append(ov)
        start += p
    overlaps, start = [], 0
    for j, p in enumerate(parts):
        gi = di = 0
        for i in range(p):
            ng, nd = n_gt[start + i], n_dt[start + i]
            overlaps.append(parted[j][gi:gi + ng, di:di + nd])
            gi, di = gi + ng, di + nd
        start += p
    return overlaps, parted, n_gt, n_dt


def prepare_data(gt_annos, dt_annos, current_class, difficulty=None, clean_data=None):
    gt_datas, dt_datas, ign_gts, ign_dets, dontcares, n_dc = [], [], [], [], [], []
    valid = 0
    for g, d in zip(gt_annos, dt_annos):
        n_valid, ign_gt, ign_det, dc = clean_data(g, d, current_class, difficulty)
        ign_gts.append(np.array(ign_gt, dtype=np.int64))
        ign_dets.append(np.array(ign_det, dtype=np.int64))
        dc = np.stack(dc, 0).astype(np.float64) if len(dc) else np.zeros((0, 4), np.float64)
        n_dc.append(dc.shape[0])
        dontcares.append(dc)
        valid += n_valid
        gt_datas.append(np.concatenate([g["bbox"], g["alpha"][..., np.newaxis]], 1))
        dt_datas.append(np.concatenate([d["bbox"], d["alpha"][..., np.newaxis], d["score"][..., np.newaxis]], 1))
    return gt_datas, dt_datas, ign_gts, ign_dets, dontcares, np.stack(n_dc, axis=0), valid


def compute_statistics_jit(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap, thresh=0,
                           compute_fp=False, compute_aos=False):
    """One frame, one class / difficulty: greedy assignment of detections to ground truths in ground-truth order.
    Without compute_fp: the highest-scoring unassigned detection above min_overlap (collects the score of every true positive:
    the recall thresholds). With compute_fp (detections below `thresh` dropped): the highest-overlap candidate, valid
    detections preferred over ignored ones; false positives = unassigned valid detections minus those inside DontCare regions
    (2-D metric); similarity = sum of (1 + cos(alpha difference)) / 2 over the true positives (orientation score).
    Returns tp, fp, fn, similarity, thresholds."""
    n_det, n_gt = dt_datas.shape[0], gt_datas.shape[0]
    scores, dt_alpha, gt_alpha = dt_datas[:, -1], dt_datas[:, 4], gt_datas[:, 4]
    assigned = np.zeros(n_det, bool)
    below = (scores < thresh) if compute_fp else np.zeros(n_det, bool)
    NONE = -10000000
    tp = fp = fn = 0
    similarity = 0
    tp_scores, deltas = [], []
    for i in range(n_gt):
        if ignored_gt[i] == -1:
            continue
        pick, best, max_ov, picked_ignored = -1, NONE, 0, False
        for j in range(n_det):
            if ignored_det[j] == -1 or assigned[j] or below[j]:
                continue
            ov = overlaps[j, i]
            if not compute_fp:
                if ov > min_overlap and scores[j] > best:
                    pick, best = j, scores[j]
            elif ov > min_overlap and (ov > max_ov or picked_ignored) and ignored_det[j] == 0:
                max_ov, pick, best, picked_ignored = ov, j, 1, False
            elif ov > min_overlap and best == NONE and ignored_det[j] == 1:
                pick, best, picked_ignored = j, 1, True
        if best == NONE:
            if ignored_gt[i] == 0:
                fn += 1
        elif ignored_gt[i] == 1 or ignored_det[pick] == 1:
            assigned[pick] = True
        else:
            tp += 1
            tp_scores.append(scores[pick])
            if compute_aos:
                deltas.append(gt_alpha[i] - dt_alpha[pick])
            assigned[pick] = True
    if compute_fp:
        candidates = ~(assigned | (ignored_det == -1) | (ignored_det == 1) | below)
        fp = int(candidates.sum())
        stuff = 0
        if metric == 0 and dc_bboxes.shape[0]:
            ov_dc = image_box_overlap(dt_datas[:, :4], dc_bboxes, 0)
            for i in range(dc_bboxes.shape[0]):
                for j in range(n_det):
                    if assigned[j] or ignored_det[j] == -1 or ignored_det[j] == 1 or below[j]:
                        continue
                    if ov_dc[j, i] > min_overlap:
                        assigned[j] = True
                        stuff += 1
        fp -= stuff
        if compute_aos:
            similarity = float(np.sum((1.0 + np.cos(np.array(deltas))) / 2.0)) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, similarity, np.array(tp_scores, dtype=np.float64)
