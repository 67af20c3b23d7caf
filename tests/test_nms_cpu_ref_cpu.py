"""The CPU NMS cores of the oracle against the reference's own det3d/ops/nms/nms_cpu.h compiled from source (boost::geometry,
not installed here, replaced by oracle/boost_shim: the vectors pin the reference's control flow, not boost's area arithmetic).
tests/golden/nms_cpu_ref.npz <- tests/golden/make_golden_nms_cpu.py."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import nms_cases as nc  # noqa: E402
from oracle import capi  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "nms_cpu_ref.npz"))
SEEDS = [int(s) for s in G["seeds"]]


def _case(s):
    c = nc.make_case(s)
    for k, v in c.items():  # the fixture stores the inputs too: the seeded generator must still produce them
        assert np.array_equal(np.asarray(v), G["c%d_%s" % (s, k)]), (s, k)
    corners = G["c%d_corners" % s]
    assert np.array_equal(capi.box2d_corners(c["boxes"][:, [0, 1, 3, 4, 6]]).astype(np.float32), corners)
    return c, corners, nc.standup_iou(corners)


@pytest.mark.parametrize("seed", SEEDS)
def test_greedy_rotated_nms_equals_the_reference_core(seed):
    """oracle/rotate_nms.c (what the device NMS is tested against) == rotate_non_max_suppression_cpu (nms_cpu.h:72-168)."""
    c, corners, su = _case(seed)
    dets = np.concatenate([c["boxes"][:, [0, 1, 3, 4, 6]], c["scores"][:, None]], 1)
    for t in nc.THRESHOLDS:
        keep, near = capi.rotate_nms_cc(dets, t)
        want = G["c%d_keep_%g" % (seed, t)]
        assert near == 0, "a pair within 1e-4 of the threshold: pick another seed"
        assert np.array_equal(np.asarray(keep, np.int32), want), (seed, t)


@pytest.mark.parametrize("seed", SEEDS)
def test_di_nms_equals_the_reference_core(seed):
    """oracle/di_nms.c == IOU_weighted_rotate_non_max_suppression_cpu (nms_cpu.h:173-384): same boxes kept in the same order, same
    labels / directions, averaged boxes and scores to float32 accuracy (the reference instantiation ran in double)."""
    c, corners, su = _case(seed)
    got = capi.di_nms_core(c["boxes"], corners, su, 0.1, c["scores"], c["iou_preds"], c["labels"], c["dirs"], c["anchors"],
                           float(c["cnt_thresh"]), nc.SIGMA_DIST_INTERVAL, nc.SIGMA_SQUARE, 0.3, int(c["centerness_c"]))
    assert list(got[4]) == list(G["c%d_di_keep" % seed]) and list(got[2]) == list(G["c%d_di_labels" % seed])
    assert list(got[3]) == list(G["c%d_di_dirs" % seed])
    if len(got[4]):
        assert np.allclose(np.asarray(got[0], np.float64).reshape(-1, 7), G["c%d_di_boxes" % seed], rtol=2e-5, atol=2e-5, equal_nan=True)
        assert np.allclose(np.asarray(got[1], np.float64), G["c%d_di_scores" % seed], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("n,thresh,eps", nc.AXIS_CASES)
def test_axis_aligned_nms_loop_equals_the_reference_core(n, thresh, eps):
    """The plain-loop restatement of nms_cpu.h:24-70 that tests/test_nms_module_gpu.py holds the device kernel to == the
    reference's non_max_suppression_cpu<float> compiled from source."""
    from test_nms_module_gpu import _axis_nms_loop
    dets, order = nc.make_axis_case(n)
    assert _axis_nms_loop(dets, order, thresh, eps) == [int(k) for k in G["axis_%d_%g_%g" % (n, thresh, eps)]]


def test_fixture_is_what_the_compiled_reference_returns_now():
    """Where /root/reference exists the header is compiled again and must reproduce the committed vectors bit for bit."""
    ref = capi.ref_nms_module()
    if ref is None:
        pytest.skip("reference sources not present: the committed fixture stands in")
    for n, t, e in nc.AXIS_CASES:
        dets, order = nc.make_axis_case(n)
        keep = ref.non_max_suppression_cpu(dets[:, :4].copy(), order, np.float32(t), np.float32(e))
        assert np.array_equal(np.asarray(keep, np.int32), G["axis_%d_%g_%g" % (n, t, e)])
    for s in SEEDS:
        c, corners, su = _case(s)
        order = np.lexsort((np.arange(len(c["scores"])), -c["scores"].astype(np.float64))).astype(np.int32)
        for t in nc.THRESHOLDS:
            keep = ref.rotate_non_max_suppression_cpu(corners.astype(np.float64), order, su.astype(np.float64), float(t))
            assert np.array_equal(np.asarray(keep, np.int32), G["c%d_keep_%g" % (s, t)])
        r = ref.IOU_weighted_rotate_non_max_suppression_cpu(
            c["boxes"].astype(np.float64), corners.astype(np.float64), su.astype(np.float64), 0.1, c["scores"].astype(np.float64),
            c["iou_preds"].astype(np.float64), c["labels"], c["dirs"], c["anchors"].astype(np.float64), float(c["cnt_thresh"]),
            nc.SIGMA_DIST_INTERVAL.astype(np.float64), nc.SIGMA_SQUARE.astype(np.float64), 0.3, int(c["centerness_c"]))
        assert np.array_equal(np.asarray(r[4], np.int32), G["c%d_di_keep" % s])
        assert np.array_equal(np.asarray(r[0], np.float64).reshape(-1, 7), G["c%d_di_boxes" % s], equal_nan=True)
