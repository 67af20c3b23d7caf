"""DI-NMS on the device (csrc/di_nms.hip = det3d/ops/nms/nms_cpu.h:173-384) through the reference's three entry
points -- box_torch_ops.rotate_weighted_nms, nms_cpu.rotate_weighted_nms_cc, the pybind-module function -- against
tests/golden/di_nms_ref.npz (the reference's wrappers run from source around the oracle core) and against the oracle directly at
the full candidate count."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

pytestmark = pytest.mark.gpu


def test_rotate_weighted_nms_matches_reference_wrappers(dev, golden_dir):
    from make_golden_di_nms import CASES, make_case
    from det3d.core.bbox import box_torch_ops as bto
    g = np.load(os.path.join(golden_dir, "di_nms_ref.npz"))
    for ci, c in enumerate(CASES):
        box, anchors, scores, iou_preds, labels, dirs = make_case(c["seed"], n=c.get("n", 260))
        args = [torch.from_numpy(a.copy()).to(dev) for a in (box, box[:, [0, 1, 3, 4, 6]], dirs, labels, scores, iou_preds, anchors)]
        res = bto.rotate_weighted_nms(*args, enable_centerness=c["cen"], centerness_pow=1, centerness_c=c["cc"], pre_max_size=c["pre"],
                                      post_max_size=None, iou_threshold=0.5, nms_cnt_thresh=c.get("cnt", 2.6))
        if "c%d_none" % ci in g.files:
            assert res is None
            continue
        b, d, l, s, sel = [r.cpu().numpy() for r in res]
        assert sel.tolist() == g["c%d_selected" % ci].tolist(), ci
        assert np.allclose(b, g["c%d_boxes" % ci].reshape(-1, 7), atol=2e-4, equal_nan=True) and np.allclose(s, g["c%d_scores" % ci], atol=1e-5), ci
        assert l.tolist() == g["c%d_labels" % ci].tolist() and d.tolist() == g["c%d_dirs" % ci].tolist()
        assert all(r.is_cuda for r in res)


def test_numpy_entry_points(dev, golden_dir):
    from make_golden_di_nms import make_case
    from det3d.ops.nms import nms_cpu
    g = np.load(os.path.join(golden_dir, "di_nms_ref.npz"))
    box, anchors, scores, iou_preds, labels, dirs = make_case(9, n=180)
    dets = np.concatenate([box[:, [0, 1, 3, 4, 6]], scores[:, None]], 1).astype(np.float32)
    for tag, an in (("cc0", None), ("cc1", anchors)):
        r = nms_cpu.rotate_weighted_nms_cc(box, dets, 0.5, iou_preds, labels.astype(np.int32), dirs.astype(np.int32), an)
        assert isinstance(r, list) and len(r) == 5 and isinstance(r[4], list)
        assert r[4] == g[tag + "_keep"].tolist()
        assert np.allclose(np.array(r[0]), g[tag + "_boxes"], atol=2e-4, equal_nan=True) and np.allclose(np.array(r[1]), g[tag + "_scores"], atol=1e-5)
        assert r[2] == g[tag + "_labels"].tolist() and r[3] == g[tag + "_dirs"].tolist()
    from det3d.ops.nms.nms import IOU_weighted_rotate_non_max_suppression_cpu as core
    assert core(np.zeros((0, 7)), np.zeros((0, 4, 2)), np.zeros((0, 0)), 0.5, np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0),
                np.zeros((1, 1)), 2.6, (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, 0) == [[], [], [], [], []]


@pytest.mark.parametrize("n,seed,cc", [(1000, 11, 0), (1024, 12, 1), (37, 13, 0), (1, 14, 1)])
def test_device_core_vs_oracle_full_size(dev, n, seed, cc):
    from make_golden_di_nms import make_case
    from det3d.core.bbox import box_np_ops
    from sessd_hip import ops
    box, anchors, scores, iou_preds, labels, dirs = make_case(seed, n=n, clusters=max(1, n // 18))
    dets = np.concatenate([box[:, [0, 1, 3, 4, 6]], scores[:, None]], 1).astype(np.float32)
    corners = box_np_ops.center_to_corner_box2d(dets[:, :2], dets[:, 2:4], dets[:, 4]).astype(np.float32)
    standup = box_np_ops.corner_to_standup_nd(corners)
    sio = box_np_ops.iou_jit(standup, standup, eps=0.0).astype(np.float32)
    want = capi.di_nms_core(box, corners, sio, 0.5, scores, iou_preds, labels.astype(np.int32), dirs.astype(np.int32),
                            anchors if cc else np.zeros((1, 1)), 2.6, (0, 20, 40, 60), (0.0009, 0.009, 0.1, 1), 0.3, cc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    b, s, l, d, k = ops.di_nms(t(box), t(corners), t(sio), t(scores), t(iou_preds), t(labels.astype(np.int32)), t(dirs.astype(np.int32)),
                               t(anchors) if cc else None)
    assert k.cpu().numpy().tolist() == want[4], (n, len(want[4]))
    if len(want[4]):
        assert np.allclose(b.cpu().numpy(), np.array(want[0]), atol=2e-4, equal_nan=True) and np.allclose(s.cpu().numpy(), np.array(want[1]), atol=1e-5)
        assert l.cpu().numpy().tolist() == want[2] and d.cpu().numpy().tolist() == want[3]
    if n >= 1000:
        assert len(want[4]) >= 20
    with pytest.raises(ValueError):
        ops.di_nms(torch.zeros((1025, 7), device=dev), torch.zeros((1025, 4, 2), device=dev), torch.zeros((1025, 1025), device=dev),
                   torch.zeros(1025, device=dev), torch.zeros(1025, device=dev), torch.zeros(1025, dtype=torch.int32, device=dev),
                   torch.zeros(1025, dtype=torch.int32, device=dev))
