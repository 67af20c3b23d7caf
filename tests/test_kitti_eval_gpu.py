"""KITTI AP evaluation with everything on the device: rotated BEV overlaps (sessd_rotate_iou_eval), 3-D overlaps
in one launch (sessd_box3d_overlap_eval), the greedy matching of every (frame, score threshold), the recall thresholds and the
tp / fp / fn / similarity sums (sessd_kitti_statistics / _thresholds / _reduce) -- against tests/golden/kitti_eval_ref.npz, the
reference's evaluation run from source on the same 24 synthetic frames, and against the host form of the same accumulation."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

pytestmark = pytest.mark.gpu


def test_official_tables_on_the_device_match_the_reference_run(dev, golden_dir):
    from make_golden_kitti_eval import make_annos
    from det3d.datasets.kitti import eval as K
    g = np.load(os.path.join(golden_dir, "kitti_eval_ref.npz"))
    gts, dts = make_annos()
    assert K.ACCUMULATE_ON_DEVICE
    res = K.get_official_eval_result(gts, dts, ["Car", "Pedestrian"])
    seen = 0
    for cls, d in res["detail"].items():
        for k, v in d.items():
            assert np.allclose(np.array(v), g["%s|%s" % (cls, k)], rtol=0, atol=1e-6), (cls, k, v, g["%s|%s" % (cls, k)])
            seen += 1
    assert seen == len([k for k in g.files if k.count("|") == 1])
    r40 = K.get_official_eval_result_v2(gts, dts, ["Car", "Pedestrian"])
    for cls, d in r40["detail"].items():
        for k, v in d.items():
            assert np.allclose(np.array(v), g["r40|%s|%s" % (cls, k)], rtol=0, atol=1e-6), (cls, k)
    coco = K.get_coco_eval_result(gts, dts, ["Car", "Pedestrian"])
    for cls, d in coco["detail"].items():
        for k, v in d.items():
            assert np.allclose(np.array(v), g["coco|%s|%s" % (cls, k)], rtol=0, atol=1e-6), (cls, k)
    min_overlaps = np.array([[[0.7, 0.5], [0.7, 0.5], [0.7, 0.5]], [[0.7, 0.5], [0.5, 0.25], [0.5, 0.25]]])
    for metric in (0, 1, 2):
        dev_r = K.eval_class_v3(gts, dts, [0, 1], [0, 1, 2], metric, min_overlaps, compute_aos=(metric == 0), on_device=True)
        host_r = K.eval_class_v3(gts, dts, [0, 1], [0, 1, 2], metric, min_overlaps, compute_aos=(metric == 0), on_device=False)
        # identical matching decisions: the device table equals the host loops exactly, and both equal the reference run
        for key in ("precision", "recall", "thresholds") + (("orientation",) if metric == 0 else ()):
            assert np.allclose(dev_r[key], host_r[key], rtol=0, atol=1e-12, equal_nan=True), (metric, key)
        assert np.allclose(dev_r["precision"], g["precision_m%d" % metric], atol=1e-6, equal_nan=True)
        assert np.allclose(dev_r["thresholds"], g["thresholds_m%d" % metric], atol=1e-6)
        if metric == 0:
            assert np.allclose(dev_r["orientation"], g["aos_m0"], atol=1e-6, equal_nan=True)


def test_fused_box3d_overlap_equals_the_two_step_form(dev):
    from det3d.datasets.utils import eval as U
    rng = np.random.RandomState(3)
    n, k = 70, 55
    def boxes(m):
        b = np.zeros((m, 7))
        b[:, 0] = rng.uniform(-10, 10, m); b[:, 1] = rng.uniform(0.5, 2.5, m); b[:, 2] = rng.uniform(5, 40, m)
        b[:, 3:6] = rng.uniform(1.2, 4.5, (m, 3)); b[:, 6] = rng.uniform(-3.2, 3.2, m)
        return b
    a, q = boxes(n), boxes(k)
    q[:20] = a[:20] + rng.normal(0, 0.15, (20, 7))   # real overlaps
    q[20:25, 1] += 50.0                               # BEV overlap but disjoint heights
    for crit in (-1, 0, 1, 2):
        for z_axis, z_center in ((1, 1.0), (2, 0.5)):
            fused = U.box3d_overlap(a, q, crit, z_axis, z_center, fused=True)
            two = U.box3d_overlap(a, q, crit, z_axis, z_center, fused=False)
            assert fused.shape == (n, k) and np.allclose(fused, two, rtol=1e-6, atol=1e-9), (crit, z_axis)
            assert (fused > 0).sum() >= 15
    assert U.box3d_overlap(a[:0], q).shape == (0, k)


def test_device_statistics_edge_cases(dev):
    """frames without detections / without ground truth / with DontCare regions / only ignored boxes, and the > 256 detections
    guard; compared with the host loop frame by frame."""
    from det3d.datasets.utils import eval as U
    from sessd_hip import ops
    from sessd_hip._lib import SessdError
    rng = np.random.RandomState(5)
    F = 9
    overlaps, gts, dts, igs, ids, dcs = [], [], [], [], [], []
    for f in range(F):
        ng = [0, 3, 5, 2, 0, 7, 1, 4, 6][f]
        nd = [4, 0, 6, 2, 0, 9, 3, 5, 8][f]
        overlaps.append(rng.uniform(0, 1, (nd, ng)) * (rng.uniform(0, 1, (nd, ng)) > 0.5))
        gts.append(np.concatenate([rng.uniform(0, 300, (ng, 4)), rng.uniform(-3, 3, (ng, 1))], 1))
        b = rng.uniform(0, 200, (nd, 2))
        dts.append(np.concatenate([b, b + rng.uniform(10, 80, (nd, 2)), rng.uniform(-3, 3, (nd, 1)), rng.uniform(0, 1, (nd, 1))], 1))
        igs.append(rng.choice([-1, 0, 0, 1], ng).astype(np.int64))
        ids.append(rng.choice([-1, 0, 0, 0, 1], nd).astype(np.int64))
        ndc = [0, 2, 0, 1, 0, 3, 0, 0, 1][f]
        c = rng.uniform(0, 150, (ndc, 2))
        dcs.append(np.concatenate([c, c + rng.uniform(40, 120, (ndc, 2))], 1))
    stat = ops.KittiStatistics(overlaps, gts, dts, igs, ids, dcs, dev)
    for metric in (0, 2):
        for mo in (0.3, 0.6):
            n_valid = int(sum((g == 0).sum() for g in igs))
            thr, pr = stat.precision_table(metric, mo, n_valid, compute_aos=(metric == 0))
            # host: same two passes
            tps = []
            for f in range(F):
                tps += U.compute_statistics_jit(overlaps[f], gts[f], dts[f], igs[f], ids[f], dcs[f], metric, mo, 0.0, False)[4].tolist()
            from det3d.datasets.kitti.eval import get_thresholds
            want_thr = np.array(get_thresholds(np.array(tps), n_valid))
            assert np.allclose(thr, want_thr, rtol=0, atol=0)
            want = np.zeros((len(want_thr), 4))
            for f in range(F):
                for t, th in enumerate(want_thr):
                    tp, fp, fn, sim, _ = U.compute_statistics_jit(overlaps[f], gts[f], dts[f], igs[f], ids[f], dcs[f], metric, mo, th,
                                                                 True, metric == 0)
                    want[t, :3] += (tp, fp, fn)
                    if sim != -1:
                        want[t, 3] += sim
            assert np.array_equal(pr[:, :3], want[:, :3]) and np.allclose(pr[:, 3], want[:, 3], rtol=0, atol=1e-12)
    big = ops.KittiStatistics([np.zeros((300, 1))], [np.zeros((1, 5))], [np.zeros((300, 6))], [np.zeros(1, np.int64)],
                              [np.zeros(300, np.int64)], [np.zeros((0, 4))], dev)
    with pytest.raises(SessdError):
        big.precision_table(0, 0.5, 1)
