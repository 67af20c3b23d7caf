"""Generates tests/golden/di_nms_ref.npz by running the REFERENCE's DI-NMS wrappers from source on CPU:

  det3d/core/bbox/box_torch_ops.py:552-621  rotate_weighted_nms   (topk, centerness damping by softmax, assembly of the outputs)
  det3d/ops/nms/nms_cpu.py:52-93            rotate_weighted_nms_cc (footprint corners, stand-up boxes, iou_jit, centerness switch)
  det3d/core/bbox/box_np_ops.py              center_to_corner_box2d, corner_to_standup_nd, iou_jit

The pybind core IOU_weighted_rotate_non_max_suppression_cpu (nms_cpu.h:173-384) needs boost::geometry and cannot be built
here: it is substituted by oracle/di_nms.c (oracle.capi.di_nms_core, same 14-argument call and 5-list return), fed with the
corners / stand-up IoU the REFERENCE code computed. So this fixture pins the wrappers around the core and the core's
restatement against itself; the core itself is pinned separately against the reference's nms_cpu.h compiled from source with a
boost::geometry stand-in (make_golden_nms_cpu.py); boost's polygon-area arithmetic stays "parity unpinned" (see oracle/di_nms.c).

    python tests/golden/make_golden_di_nms.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
sys.path.insert(0, HERE)


def make_case(seed, n=260, clusters=14):
    """n candidate boxes in `clusters` groups around 'true' objects (the situation DI-NMS is made for), with anchors, IoU
    predictions in (0.3, 1), two labels, direction bits; scores in (0.3, 0.99) all distinct."""
    rng = np.random.RandomState(seed)
    centers = np.stack([rng.uniform(2, 68, clusters), rng.uniform(-38, 38, clusters)], 1)
    which = rng.randint(0, clusters, n)
    box = np.zeros((n, 7), np.float32)
    box[:, :2] = centers[which] + rng.normal(0, 0.25, (n, 2))
    box[:, 2] = rng.uniform(-1.2, -0.8, n)
    box[:, 3:6] = np.array([1.6, 3.9, 1.56]) + rng.normal(0, 0.08, (n, 3))
    base_r = rng.uniform(-3.1, 3.1, clusters)
    box[:, 6] = base_r[which] + rng.normal(0, 0.06, n)
    anchors = box.copy()
    anchors[:, :2] += rng.normal(0, 0.3, (n, 2))
    scores = rng.permutation(np.linspace(0.3, 0.99, n)).astype(np.float32)
    iou_preds = rng.uniform(0.3, 1.0, n).astype(np.float32)
    labels = (which % 5 == 0).astype(np.int64)          # a few clusters carry the second label
    dirs = rng.randint(0, 2, n).astype(np.int64)
    return box, anchors.astype(np.float32), scores, iou_preds, labels, dirs


# pre_max_size is always given: without it the reference's own `indices[keep]` (box_torch_ops.py:621) is unbound
CASES = [dict(seed=1, pre=1000, cc=False, cen=True), dict(seed=2, pre=200, cc=True, cen=True), dict(seed=3, pre=120, cc=False, cen=False),
         dict(seed=4, pre=1000, cc=False, cen=True, cnt=0.8), dict(seed=5, pre=10, cc=True, cen=True, n=1), dict(seed=6, pre=50, cc=False, cen=True, n=0)]


def main():
    assert os.path.isdir("/root/reference")
    warnings.filterwarnings("ignore")
    from oracle import capi
    import make_golden_head_loss as HL
    HL.install(capi)
    mod, load_as = HL.mod, HL.load_as
    mod("det3d.ops"); mod("det3d.ops.nms")
    mod("det3d.ops.nms.nms", non_max_suppression_cpu=None, rotate_non_max_suppression_cpu=None,
        IOU_weighted_rotate_non_max_suppression_cpu=capi.di_nms_core)
    nms_cpu = load_as("det3d/ops/nms/nms_cpu.py", "det3d.ops.nms.nms_cpu")
    nms_cpu.IOU_weighted_rotate_non_max_suppression_cpu = capi.di_nms_core
    bto = sys.modules["det3d.core.bbox.box_torch_ops"]
    bto.rotate_weighted_nms_cc = nms_cpu.rotate_weighted_nms_cc
    out = {}
    for ci, c in enumerate(CASES):
        box, anchors, scores, iou_preds, labels, dirs = make_case(c["seed"], n=c.get("n", 260))
        args = [torch.from_numpy(a.copy()) for a in (box, box[:, [0, 1, 3, 4, 6]], dirs, labels, scores, iou_preds, anchors)]
        res = bto.rotate_weighted_nms(*args, enable_centerness=c["cen"], centerness_pow=1, centerness_c=c["cc"], pre_max_size=c["pre"],
                                      post_max_size=None, iou_threshold=0.5, nms_cnt_thresh=c.get("cnt", 2.6))
        if res is None:   # the reference's empty branch assigns five empty arrays and falls off the end of the function
            out["c%d_none" % ci] = np.array([1])
            print("case", ci, c, "-> None")
            continue
        names = ("boxes", "dirs", "labels", "scores", "selected")
        for nm, v in zip(names, res):
            v = v.numpy() if torch.is_tensor(v) else np.asarray(v)
            out["c%d_%s" % (ci, nm)] = v
        print("case", ci, c, "kept", len(out["c%d_selected" % ci]))
    # the numpy-level entry point on its own (no centerness inside the core, then with it)
    box, anchors, scores, iou_preds, labels, dirs = make_case(9, n=180)
    dets = np.concatenate([box[:, [0, 1, 3, 4, 6]], scores[:, None]], 1).astype(np.float32)
    for tag, an in (("cc0", None), ("cc1", anchors)):
        r = nms_cpu.rotate_weighted_nms_cc(box, dets, 0.5, iou_preds, labels.astype(np.int32), dirs.astype(np.int32), an)
        for nm, v in zip(("boxes", "scores", "labels", "dirs", "keep"), r):
            out["%s_%s" % (tag, nm)] = np.asarray(v)
        print(tag, "kept", len(r[4]))
    np.savez_compressed(os.path.join(HERE, "di_nms_ref.npz"), **out)
    print("wrote", os.path.join(HERE, "di_nms_ref.npz"), os.path.getsize(os.path.join(HERE, "di_nms_ref.npz")), "bytes")


if __name__ == "__main__":
    main()
