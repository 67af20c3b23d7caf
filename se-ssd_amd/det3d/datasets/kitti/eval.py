"""mirrors det3d/datasets/kitti/eval.py: KITTI average precision (2-D bbox / BEV / 3-D / orientation) as the reference
computes it for the "car 3D AP@0.7" headline of SE-SSD -- get_official_eval_result :467-569, do_eval_v3 :395-421,
eval_class_v3 :174-319, fused_compute_statistics :121-171, clean_data :40-108, get_thresholds :18-37, get_mAP :330-340.
The rotated-box overlaps (det3d.datasets.utils.eval), the greedy matching of every (frame, score threshold), the recall
thresholds and the tp / fp / fn / similarity sums run on the device (sessd_kitti_* in csrc/kitti_eval.hip); difficulty
filtering (strings, a few boxes per frame) and the final 41-point table arithmetic stay on the host. The host loops of the
reference's structure remain as the second implementation (on_device=False). Pinned by tests/golden/kitti_eval_ref.npz."""
import io as sysio

import numpy as np

from det3d.datasets.utils.eval import calculate_iou_partly, compute_statistics_jit, get_split_parts, prepare_data

CLASS_NAMES = ["car", "pedestrian", "bicycle", "truck", "bus", "trailer", "construction_vehicle", "motorcycle", "barrier",
               "traffic_cone", "cyclist"]
MIN_HEIGHT = [40, 25, 25]          # 2-D box height in pixels: easy / moderate / hard
MAX_OCCLUSION = [0, 1, 2]
MAX_TRUNCATION = [0.15, 0.3, 0.5]


def get_thresholds(scores, num_gt, num_sample_pts=41):
    """Scores at which the recall crosses the next of `num_sample_pts` equally spaced levels (:18-37)."""
    scores = np.sort(scores)[::-1]
    out, current = [], 0.0
    for i, s in enumerate(scores):
        left = (i + 1) / num_gt
        right = (i + 2) / num_gt if i < len(scores) - 1 else left
        if (right - current) < (current - left) and i < len(scores) - 1:
            continue
        out.append(s)
        current += 1 / (num_sample_pts - 1.0)
    return out


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """Per ground truth: 0 = counts, 1 = neutral (neighbouring class -- Van for Car, Person_sitting for Pedestrian -- or too
    hard for this difficulty), -1 = other class; DontCare 2-D boxes; per detection: 0 = counts, 1 = too small, -1 = other class."""
    cls = CLASS_NAMES[current_class].lower()
    ignored_gt, ignored_dt, dc = [], [], []
    valid = 0
    for i in range(len(gt_anno["name"])):
        bbox = gt_anno["bbox"][i]
        name = gt_anno["name"][i].lower()
        if name == cls:
            kind = 1
        elif (cls == "pedestrian" and name == "person_sitting") or (cls == "car" and name == "van"):
            kind = 0
        else:
            kind = -1
        hard = (gt_anno["occluded"][i] > MAX_OCCLUSION[difficulty] or gt_anno["truncated"][i] > MAX_TRUNCATION[difficulty]
                or (bbox[3] - bbox[1]) <= MIN_HEIGHT[difficulty])
        if kind == 1 and not hard:
            ignored_gt.append(0)
            valid += 1
        elif kind == 0 or (hard and kind == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
        if gt_anno["name"][i] in ("DontCare", "ignore"):
            dc.append(bbox)
    for i in range(len(dt_anno["name"])):
        height = abs(dt_anno["bbox"][i, 3] - dt_anno["bbox"][i, 1])
        if height < MIN_HEIGHT[difficulty]:
            ignored_dt.append(1)
        elif dt_anno["name"][i].lower() == cls:
            ignored_dt.append(0)
        else:
            ignored_dt.append(-1)
    return valid, ignored_gt, ignored_dt, dc


def fused_compute_statistics(overlaps, pr, gt_nums, dt_nums, dc_nums, gt_datas, dt_datas, dontcares, ignored_gts, ignored_dets,
                             metric, min_overlap, thresholds, compute_aos=False):
    """Accumulate (tp, fp, fn, similarity) of a group of frames into pr[len(thresholds), 4] (:121-171)."""
    g = d = c = 0
    for i in range(gt_nums.shape[0]):
        sl_g, sl_d, sl_c = slice(g, g + gt_nums[i]), slice(d, d + dt_nums[i]), slice(c, c + dc_nums[i])
        for t, thresh in enumerate(thresholds):
            tp, fp, fn, sim, _ = compute_statistics_jit(overlaps[sl_d, sl_g], gt_datas[sl_g], dt_datas[sl_d], ignored_gts[sl_g],
                                                        ignored_dets[sl_d], dontcares[sl_c], metric, min_overlap=min_overlap,
                                                        thresh=thresh, compute_fp=True, compute_aos=compute_aos)
            pr[t, 0] += tp
            pr[t, 1] += fp
            pr[t, 2] += fn
            if sim != -1:
                pr[t, 3] += sim
        g, d, c = g + gt_nums[i], d + dt_nums[i], c + dc_nums[i]


DEVICE_MAX_DETECTIONS_PER_FRAME = 256  # MAXDET of csrc/kitti_eval.hip
ACCUMULATE_ON_DEVICE = True  # False: the host loops below (the reference's structure; what the CPU tests pin to the golden)


def eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, z_axis=1,
                  z_center=1.0, num_parts=50, on_device=None):
    """Precision at 41 recall levels per (class, difficulty, overlap threshold) (:174-319). metric 0 bbox / 1 BEV / 2 3-D;
    min_overlaps [num_minoverlap, metric, class]. on_device (default: ACCUMULATE_ON_DEVICE and a GPU is present): the greedy
    matching of every (frame, score threshold) and the tp / fp / fn / similarity sums run on the MI355X
    (sessd_kitti_statistics / _thresholds / _reduce), only the (<= 41, 4) table per configuration comes back."""
    if on_device is None:
        import torch
        on_device = ACCUMULATE_ON_DEVICE and torch.cuda.is_available()
    assert len(gt_annos) == len(dt_annos)
    parts = [p for p in get_split_parts(len(gt_annos), num_parts) if p != 0]
    overlaps, parted, n_dt, n_gt = calculate_iou_partly(dt_annos, gt_annos, metric, num_parts, z_axis=z_axis, z_center=z_center)
    PTS = 41
    shape = [len(current_classes), len(difficultys), len(min_overlaps), PTS]
    precision, recall, aos, all_thr = np.zeros(shape), np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(current_classes):
        for l, diff in enumerate(difficultys):
            gt_datas, dt_datas, ign_gts, ign_dets, dontcares, n_dc, n_valid = prepare_data(gt_annos, dt_annos, cls, difficulty=diff,
                                                                                         clean_data=clean_data)
            stat = None
            # the device kernels keep a frame's detections in fixed 256-entry bit masks (MAXDET, csrc/kitti_eval.hip): a
            # (class, difficulty) with a busier frame -- low score thresholds, many classes -- takes the host loops below,
            # which have no such limit (the reference's structure)
            if on_device and max([d.shape[0] for d in dt_datas] + [0]) <= DEVICE_MAX_DETECTIONS_PER_FRAME:
                import torch
                from sessd_hip import ops
                stat = ops.KittiStatistics(overlaps, gt_datas, dt_datas, ign_gts, ign_dets, dontcares,
                                           torch.device("cuda", torch.cuda.current_device()))
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                if stat is not None:
                    thresholds, pr = stat.precision_table(metric, float(min_overlap), int(n_valid), compute_aos, PTS)
                    n = len(thresholds)
                    all_thr[m, l, k, :n] = thresholds
                    with np.errstate(invalid="ignore", divide="ignore"):
                        precision[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                        if compute_aos:
                            aos[m, l, k, :n] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                    for i in range(n):
                        precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                        if compute_aos:
                            aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
                    continue
                tp_scores = []
                for i in range(len(gt_annos)):
                    tp_scores += compute_statistics_jit(overlaps[i], gt_datas[i], dt_datas[i], ign_gts[i], ign_dets[i], dontcares[i],
                                                        metric, min_overlap=min_overlap, thresh=0.0, compute_fp=False)[4].tolist()
                thresholds = np.array(get_thresholds(np.array(tp_scores), n_valid))
                all_thr[m, l, k, :len(thresholds)] = thresholds
                pr = np.zeros([len(thresholds), 4])
                idx = 0
                for j, p in enumerate(parts):
                    cat = lambda lst: np.concatenate(lst[idx:idx + p], 0)
                    fused_compute_statistics(parted[j], pr, n_gt[idx:idx + p], n_dt[idx:idx + p], n_dc[idx:idx + p], cat(gt_datas),
                                             cat(dt_datas), cat(dontcares), cat(ign_gts), cat(ign_dets), metric,
                                             min_overlap=min_overlap, thresholds=thresholds, compute_aos=compute_aos)
                    idx += p
                n = len(thresholds)
                with np.errstate(invalid="ignore", divide="ignore"):
                    precision[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, l, k, :n] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                for i in range(n):  # monotone envelope (over all 41 slots, like the reference)
                    precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
    # `recall` stays zero as in the reference (:292 is commented out there)
    return {"recall": recall, "precision": precision, "orientation": aos, "thresholds": all_thr, "min_overlaps": min_overlaps}


def get_mAP(prec):
    """11-point interpolated AP in percent (:330-333)."""
    return prec[..., ::4].sum(-1) / 11 * 100


def get_mAP_v2(prec):
    """40-point AP (R40) in percent (:336-340)."""
    return prec[..., 1:].sum(-1) / 40 * 100


def do_eval_v3(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, difficultys=(0, 1, 2), z_axis=1, z_center=1.0):
    return {name: eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos, z_axis=z_axis,
                                z_center=z_center) for metric, name in enumerate(("bbox", "bev", "3d"))}


def print_str(value, *arg, sstream=None):
    if sstream is None:
        sstream = sysio.StringIO()
    sstream.truncate(0)
    sstream.seek(0)
    print(value, *arg, file=sstream)
    return sstream.getvalue()


def _class_ids(current_classes):
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [CLASS_NAMES.index(c.lower()) if isinstance(c, str) else c for c in current_classes]


def _has_alpha(dt_annos):
    for anno in dt_annos:
        if anno["alpha"].shape[0] != 0:
            return bool(anno["alpha"][0] != -10)
    return False


def _official(gt_annos, dt_annos, current_classes, difficultys, z_axis, z_center, ap):
    strict = np.array([[0.7, 0.5, 0.5, 0.7, 0.7, 0.7, 0.7, 0.5, 0.5, 0.5, 0.5]] * 3)
    relaxed = np.array([[0.7, 0.5, 0.5, 0.7, 0.7, 0.7, 0.7, 0.5, 0.25, 0.25, 0.5],
                        [0.5, 0.25, 0.25, 0.5, 0.5, 0.5, 0.5, 0.25, 0.25, 0.25, 0.25],
                        [0.5, 0.25, 0.25, 0.5, 0.5, 0.5, 0.5, 0.25, 0.25, 0.25, 0.25]])
    classes = _class_ids(current_classes)
    min_overlaps = np.stack([strict, relaxed], axis=0)[:, :, classes]
    compute_aos = _has_alpha(dt_annos)
    metrics = do_eval_v3(gt_annos, dt_annos, classes, min_overlaps, compute_aos, difficultys, z_axis=z_axis, z_center=z_center)
    result, detail = "", {}
    fmt = lambda v: ", ".join("%.2f" % x for x in v)
    for j, cls in enumerate(classes):
        name = CLASS_NAMES[cls]
        detail[name] = {}
        for i in range(min_overlaps.shape[0]):
            val = {k: ap(metrics[k]["precision"][j, :, i]) for k in ("bbox", "bev", "3d")}
            detail[name]["bbox@%.2f" % min_overlaps[i, 0, j]] = val["bbox"].tolist()
            detail[name]["bev@%.2f" % min_overlaps[i, 1, j]] = val["bev"].tolist()
            detail[name]["3d@%.2f" % min_overlaps[i, 2, j]] = val["3d"].tolist()
            result += print_str("%s AP(Average Precision)@%.2f, %.2f, %.2f:" % ((name,) + tuple(min_overlaps[i, :, j])))
            result += print_str("bbox AP:" + fmt(val["bbox"]))
            result += print_str("bev  AP:" + fmt(val["bev"]))
            result += print_str("3d   AP:" + fmt(val["3d"]))
            if compute_aos:
                a = ap(metrics["bbox"]["orientation"][j, :, i])
                detail[name]["aos"] = a.tolist()
                result += print_str("aos  AP:" + fmt(a))
    return {"result": result, "detail": detail}


def get_official_eval_result(gt_annos, dt_annos, current_classes, difficultys=[0, 1, 2], z_axis=1, z_center=1.0):
    """{'result': the printed table, 'detail': {class: {'bbox@0.70': [easy, moderate, hard], 'bev@..', '3d@..', 'aos'}}} with
    the official overlap thresholds and their relaxed variant, 11-point AP (:467-569)."""
    return _official(gt_annos, dt_annos, current_classes, difficultys, z_axis, z_center, get_mAP)


def get_official_eval_result_v2(gt_annos, dt_annos, current_classes, difficultys=[0, 1, 2], z_axis=1, z_center=1.0):
    """The same table with the 40-point AP (R40) (:571-672)."""
    return _official(gt_annos, dt_annos, current_classes, difficultys, z_axis, z_center, get_mAP_v2)


def do_eval_v2(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, difficultys=(0, 1, 2), z_axis=1, z_center=1.0):
    """(mAP_bbox, mAP_bev, mAP_3d, mAP_aos), each [class, difficulty, overlap] (:343-392)."""
    r = eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps, compute_aos, z_axis=z_axis, z_center=z_center)
    bbox, aos = get_mAP(r["precision"]), (get_mAP(r["orientation"]) if compute_aos else None)
    bev = get_mAP(eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps, z_axis=z_axis, z_center=z_center)["precision"])
    d3 = get_mAP(eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps, z_axis=z_axis, z_center=z_center)["precision"])
    return bbox, bev, d3, aos


def do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos, z_axis=1, z_center=1.0):
    """AP averaged over 10 overlap thresholds per (metric, class): overlap_ranges [start/stop/num, metric, class] (:424-455)."""
    min_overlaps = np.zeros([10, *overlap_ranges.shape[1:]])
    for i in range(overlap_ranges.shape[1]):
        for j in range(overlap_ranges.shape[2]):
            start, stop, num = overlap_ranges[:, i, j]
            min_overlaps[:, i, j] = np.linspace(start, stop, int(num))
    bbox, bev, d3, aos = do_eval_v2(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, z_axis=z_axis, z_center=z_center)
    return bbox.mean(-1), bev.mean(-1), d3.mean(-1), (None if aos is None else aos.mean(-1))


def get_coco_eval_result(gt_annos, dt_annos, current_classes, z_axis=1, z_center=1.0):
    """COCO-style AP over overlap 0.50:0.05:0.95 (vehicles) / 0.25:0.05:0.70 (small classes) (:675-790)."""
    wide = {0, 3, 4, 5, 6}
    classes = _class_ids(current_classes)
    ranges = np.zeros([3, 3, len(classes)])
    for i, c in enumerate(classes):
        ranges[:, :, i] = np.array([0.5, 0.95, 10] if c in wide else [0.25, 0.7, 10])[:, np.newaxis]
    compute_aos = _has_alpha(dt_annos)
    bbox, bev, d3, aos = do_coco_style_eval(gt_annos, dt_annos, classes, ranges, compute_aos, z_axis=z_axis, z_center=z_center)
    result, detail = "", {}
    for j, c in enumerate(classes):
        name = CLASS_NAMES[c]
        start, stop, num = ([0.5, 0.95, 10] if c in wide else [0.25, 0.7, 10])
        result += print_str("%s coco AP@%.2f:%.2f:%.2f:" % (name, start, (stop - start) / (num - 1), stop))
        result += print_str("bbox AP:%.2f, %.2f, %.2f" % tuple(bbox[j]))
        result += print_str("bev  AP:%.2f, %.2f, %.2f" % tuple(bev[j]))
        result += print_str("3d   AP:%.2f, %.2f, %.2f" % tuple(d3[j]))
        detail[name] = {"bbox": bbox[j].tolist(), "bev": bev[j].tolist(), "3d": d3[j].tolist()}
        if compute_aos:
            detail[name]["aos"] = aos[j].tolist()
            result += print_str("aos  AP:%.2f, %.2f, %.2f" % tuple(aos[j]))
    return {"result": result, "detail": detail}
