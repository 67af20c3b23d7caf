"""Generates tests/golden/nms_cpu_ref.npz by running the REFERENCE'S OWN det3d/ops/nms/nms_cpu.h, compiled from source
(oracle/build.py build_ref_nms: the header is compiled where it lies; <boost/geometry.hpp>, not installed here, is resolved to
oracle/boost_shim -- convex clipping + shoelace in double, see that header). What the vectors pin is the reference's control
flow: the axis-aligned NMS (nms_cpu.h:24-70; no polygon arithmetic: pinned completely), the greedy rotated NMS (:72-168) and DI-NMS (IOU_weighted_rotate_non_max_suppression_cpu, :173-384: score
normalisation, centerness, pick / count / weighted average / suppress / recover). boost's own area arithmetic stays unpinned.
Run in the build container (needs /root/reference): python tests/golden/make_golden_nms_cpu.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import capi  # noqa: E402
import nms_cases as nc  # noqa: E402

ref = capi.ref_nms_module()
assert ref is not None, "the reference header could not be compiled (needs /root/reference)"
out = {}
SEEDS = [s for s in range(1, 18) if s != 6]  # seed 6 has an IoU within 1e-4 of a threshold (a decision boost itself could take either way)
for s in SEEDS:
    c = nc.make_case(s)
    corners = capi.box2d_corners(c["boxes"][:, [0, 1, 3, 4, 6]]).astype(np.float32)  # center_to_corner_box2d order (clockwise)
    su = nc.standup_iou(corners)
    for k, v in c.items():
        out["c%d_%s" % (s, k)] = v
    out["c%d_corners" % s] = corners
    order = np.lexsort((np.arange(len(c["scores"])), -c["scores"].astype(np.float64))).astype(np.int32)
    for t in nc.THRESHOLDS:
        keep = ref.rotate_non_max_suppression_cpu(corners.astype(np.float64), order, su.astype(np.float64), float(t))
        out["c%d_keep_%g" % (s, t)] = np.asarray(keep, np.int32)
    r = ref.IOU_weighted_rotate_non_max_suppression_cpu(
        c["boxes"].astype(np.float64), corners.astype(np.float64), su.astype(np.float64), 0.1, c["scores"].astype(np.float64),
        c["iou_preds"].astype(np.float64), c["labels"], c["dirs"], c["anchors"].astype(np.float64), float(c["cnt_thresh"]),
        nc.SIGMA_DIST_INTERVAL.astype(np.float64), nc.SIGMA_SQUARE.astype(np.float64), 0.3, int(c["centerness_c"]))
    out["c%d_di_boxes" % s] = np.asarray(r[0], np.float64).reshape(-1, 7)
    out["c%d_di_scores" % s] = np.asarray(r[1], np.float64)
    out["c%d_di_labels" % s] = np.asarray(r[2], np.int32)
    out["c%d_di_dirs" % s] = np.asarray(r[3], np.int32)
    out["c%d_di_keep" % s] = np.asarray(r[4], np.int32)
for n, t, e in nc.AXIS_CASES:  # axis-aligned NMS with the +eps convention (float32 instantiation, as the wrappers reach it)
    dets, order = nc.make_axis_case(n)
    out["axis_%d_%g_%g" % (n, t, e)] = np.asarray(ref.non_max_suppression_cpu(dets[:, :4].copy(), order, np.float32(t), np.float32(e)), np.int32)
out["seeds"] = np.asarray(SEEDS, np.int32)
np.savez_compressed(os.path.join(HERE, "nms_cpu_ref.npz"), **out)
print("wrote nms_cpu_ref.npz:", sum(len(out["c%d_di_keep" % s]) for s in SEEDS), "DI-NMS boxes,",
      sum(len(out["c%d_keep_0.3" % s]) for s in SEEDS), "NMS keeps at 0.3")
