"""Anchor target assignment on the device (sessd_assign_targets; det3d.datasets.pipelines.AssignTarget) vs the CPU oracle
(oracle/assign_target.py) and the reference's own create_target_np run from source (tests/golden/assign_ref.npz).
labels / foreground set / assigned ground-truth ids must be IDENTICAL (integer work); regression targets within 1e-6
(float32 log / sqrt of the device library vs numpy)."""
import os

import numpy as np
import pytest
import torch

from oracle import assign_target as oat, postprocess as pp
from sessd_hip import ops

pytestmark = pytest.mark.gpu


def test_kernel_vs_reference_golden_and_oracle(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "assign_ref.npz"))
    anchors = pp.create_anchors_3d_range().reshape(-1, 7).astype(np.float32)
    A = torch.from_numpy(anchors).to(dev)
    for c in "abcd":
        gt = g[c + "_gt"]
        r = ops.assign_targets(A, torch.from_numpy(gt).to(dev))
        labels = r["labels"].cpu().numpy()
        assert np.array_equal(labels, g[c + "_labels"]), c
        pos = np.nonzero(labels > 0)[0]
        assert np.array_equal(pos, g[c + "_pos"])
        t = r["bbox_targets"].cpu().numpy()
        if len(pos):
            assert np.abs(t[pos] - g[c + "_targets_pos"]).max() < 1e-6
        assert np.all(t[labels <= 0] == 0)
        gid = r["gt_id"].cpu().numpy()
        assert np.array_equal(gid[gid >= 0], g[c + "_gt_id"])
        assert float(r["bbox_outside_weights"].sum()) == float(g[c + "_weights_sum"])
        want = oat.assign(anchors, gt)
        assert np.array_equal(labels, want["labels"]) and np.abs(t - want["bbox_targets"]).max() < 1e-6


def test_pipeline_stage_contract(dev):
    from det3d.datasets.pipelines import AssignTarget
    from det3d.torchie import Config
    from sessd_hip import configs
    cfg = Config.fromfile(configs.REFERENCE_CONFIG) if hasattr(configs, "REFERENCE_CONFIG") and os.path.exists(configs.REFERENCE_CONFIG) else None
    assigner = dict(target_assigner=dict(anchor_generators=[dict(type="anchor_generator_range", sizes=[1.6, 3.9, 1.56],
                    anchor_ranges=[0, -40.0, -1.0, 70.4, 40.0, -1.0], rotations=[0, 1.57], matched_threshold=0.6, unmatched_threshold=0.45,
                    class_name="Car")]), out_size_factor=8, enable_similar_type=True) if cfg is None else cfg.train_cfg["assigner"]
    stage = AssignTarget(cfg=assigner)
    rng = np.random.RandomState(1)
    gt = np.zeros((6, 7), np.float32)
    gt[:, 0] = rng.uniform(5, 60, 6); gt[:, 1] = rng.uniform(-30, 30, 6); gt[:, 2] = -1.0
    gt[:, 3:6] = [1.6, 3.9, 1.5]; gt[:, 6] = rng.uniform(-6, 6, 6)   # yaw outside [-pi, pi): folded by the stage
    ann = lambda: dict(gt_boxes=gt.copy(), gt_classes=np.array([1, 1, 2, 1, 3, 1], np.int32), gt_names=np.array(["Car", "Car", "Van", "Car", "Cyclist", "Car"]))
    res = dict(mode="train", labeled=True, lidar=dict(annotations=ann(), annotations_raw=ann()))
    res, _ = stage(res, None)
    T = res["lidar"]["targets"]
    assert T["anchors"][0].shape == (70400, 7) and T["labels"][0].shape == (70400,) and T["reg_targets"][0].shape == (70400, 7)
    kept = gt[[0, 1, 2, 3, 5]].copy()                                  # classes 1 and 2 (enable_similar_type)
    kept[:, 6] = kept[:, 6] - np.floor(kept[:, 6] / (2 * np.pi) + 0.5) * (2 * np.pi)
    want = oat.assign(T["anchors"][0], kept)
    assert np.array_equal(T["labels"][0], want["labels"]) and np.abs(T["reg_targets"][0] - want["bbox_targets"]).max() < 1e-6
    assert np.array_equal(T["positive_gt_id"][0][0], want["positive_gt_id"]) and float(T["reg_weights"][0].sum()) == float((want["labels"] > 0).sum())
    assert np.array_equal(res["lidar"]["targets_raw"]["labels"][0], T["labels"][0])
    val = stage(dict(mode="val", lidar=dict()), None)[0]
    assert list(val["lidar"]["targets"].keys()) == ["anchors"]
