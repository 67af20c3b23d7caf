"""Generates tests/golden/assign_ref.npz by running the REFERENCE's target assignment from source on CPU:
    det3d/core/anchor/target_ops_v3.py   create_target_np :11-137
    det3d/core/bbox/region_similarity.py NearestIouSimilarity :75-98 (rbbox2d_to_near_bbox + iou_jit(eps=0), box_np_ops.py:354-366,1008-1046)
    det3d/core/bbox/box_np_ops.py        second_box_encode :52-110 (what GroundBox3dCoder.encode calls)
as det3d/core/anchor/target_assigner.py:68-136 (assign_v2) wires them for the car anchors of config.py:82-100
(matched 0.6 / unmatched 0.45, sample_size 512, no positive fraction). numba is stubbed (the kernels run as Python loops).
Run in the build container only:  python tests/golden/make_golden_assign.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
sys.path.insert(0, HERE)


def main():
    import make_golden as MG
    assert os.path.isdir(MG.REF)
    MG.install_stubs()
    import types
    np.bool = bool  # removed alias still used by the reference
    MG.load_ref("det3d/core/bbox/geometry.py", "det3d.core.bbox.geometry")
    bnp = MG.load_ref("det3d/core/bbox/box_np_ops.py", "det3d.core.bbox.box_np_ops")
    sys.modules["det3d.core.bbox"].box_np_ops = bnp
    ops = MG.load_ref("det3d/core/anchor/target_ops_v3.py", "ref_target_ops")
    from oracle import postprocess as pp
    anchors = pp.create_anchors_3d_range().reshape(-1, 7).astype(np.float32)   # pinned to the reference generator elsewhere
    N = anchors.shape[0]
    rng = np.random.RandomState(0)

    def similarity(a, g):  # NearestIouSimilarity._compare on [x, y, w, l, r] (target_assigner.py:79-82)
        return bnp.iou_jit(bnp.rbbox2d_to_near_bbox(a[:, [0, 1, 3, 4, -1]]), bnp.rbbox2d_to_near_bbox(g[:, [0, 1, 3, 4, -1]]), eps=0.0)

    def encode(boxes, anc):
        return bnp.second_box_encode(boxes, anc, False, False)

    def gts(m, far=False):
        b = np.zeros((m, 7), np.float32)
        b[:, 0] = rng.uniform(2, 68, m); b[:, 1] = rng.uniform(-38, 38, m); b[:, 2] = rng.uniform(-1.6, -0.4, m)
        b[:, 3] = rng.uniform(1.4, 1.9, m); b[:, 4] = rng.uniform(3.2, 4.6, m); b[:, 5] = rng.uniform(1.3, 1.8, m)
        b[:, 6] = rng.uniform(-np.pi, np.pi, m)
        if far and m:
            b[-1, :2] = [150.0, 150.0]   # overlaps no anchor: the empty_gt_mask branch (:65-67)
        return b

    out = dict(anchors_checksum=np.array(anchors.sum(0)))
    for name, g in (("a", gts(12)), ("b", gts(5, far=True)), ("c", gts(0)), ("d", gts(30))):
        t = ops.create_target_np(anchors, g, similarity, encode, prune_anchor_fn=None, gt_classes=np.ones(len(g), np.int32),
                                 matched_threshold=np.full(N, 0.6, np.float32), unmatched_threshold=np.full(N, 0.45, np.float32),
                                 positive_fraction=None, rpn_batch_size=512, norm_by_num_examples=False, box_code_size=7)
        out[name + "_gt"] = g
        out[name + "_labels"] = t["labels"]
        pos = np.nonzero(t["labels"] > 0)[0]
        out[name + "_pos"] = pos
        out[name + "_targets_pos"] = t["bbox_targets"][pos]
        assert np.all(t["bbox_targets"][t["labels"] <= 0] == 0)
        out[name + "_weights_sum"] = np.array(t["bbox_outside_weights"].sum())
        out[name + "_gt_id"] = t["positive_gt_id"]
        print(name, "gt", len(g), "pos", len(pos), "neg", int((t["labels"] == 0).sum()), "ignore", int((t["labels"] < 0).sum()))
    np.savez_compressed(os.path.join(HERE, "assign_ref.npz"), **out)


if __name__ == "__main__":
    main()
