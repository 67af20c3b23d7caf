"""GPU twins of tests/test_forward_golden_cpu.py: the det3d-mirror modules (HIP kernels underneath) against what the
REFERENCE's own classes computed when run from source (tests/golden/forward_ref.npz, make_golden_forward.py):
SSFA.forward, Head.forward, VoxelFeatureExtractorV3.forward, MultiGroupHead.predict. Tolerances: features 2e-4 * max|ref|
(float32 sums in a different order; Winograd layers included), boxes 1e-3, scores 1e-3 relative, selection identical."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import forward_cases as FC  # noqa: E402
from test_forward_golden_cpu import G, head_state, predict_inputs, ssfa_state  # noqa: E402,F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(dev):
    from det3d.models import build_detector
    from sessd_hip import configs
    m = build_detector(configs.kitti_car_model(), train_cfg=None, test_cfg=configs.TEST_CFG)
    return m.to(dev).eval()


def test_ssfa_module_equals_reference_class(G, model, dev):
    neck = model.neck
    sd = ssfa_state(G)
    assert sorted(neck.state_dict()) == sorted(sd)  # the reference's state_dict names
    neck.load_state_dict(sd)
    neck.eval()
    x = FC.ssfa_input().to(dev)
    with torch.no_grad():
        got = neck(x).cpu().numpy()
    ref = G["ssfa_eval"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


def test_head_module_equals_reference_class(G, model, dev):
    task = model.bbox_head.tasks[0]
    task.load_state_dict(head_state(G))
    with torch.no_grad():
        got = model.bbox_head(FC.head_input().to(dev))[0]
    for k in ("box_preds", "cls_preds", "dir_cls_preds", "iou_preds"):
        ref = G["head_" + k]
        assert tuple(got[k].shape) == ref.shape
        assert np.abs(got[k].cpu().numpy() - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


def test_vfe_module_and_kernel_equal_reference_class(G, model, dev):
    vox, num = FC.vfe_case()
    ref = G["vfe_mean"]
    got = model.reader(torch.from_numpy(vox).to(dev), torch.from_numpy(num).to(dev)).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case,seed", [("a", 21), ("b", 22)])
def test_predict_module_equals_reference_method(G, model, dev, case, seed):
    from sessd_hip import configs
    pc, anchors = predict_inputs(G, case, seed)
    B = pc["box_preds"].shape[0]
    frustum = torch.from_numpy(np.broadcast_to(G["predict_frustum"], (B,) + G["predict_frustum"].shape).copy())
    example = dict(anchors=[torch.from_numpy(np.broadcast_to(anchors, (B,) + anchors.shape).copy()).to(dev)],
                   metadata=[dict(token=str(i)) for i in range(B)], calib=dict(frustum=frustum.to(dev)))
    preds = [{k: torch.from_numpy(v).to(dev) for k, v in pc.items()}]
    rets = model.bbox_head.predict(example, preds, configs.TEST_CFG)
    assert len(rets) == B
    for b, r in enumerate(rets):
        rb, rs, rl = (G["predict_%s%d_%s" % (case, b, k)] for k in ("boxes", "scores", "labels"))
        s = r["scores"].cpu().numpy()
        assert s.shape == rs.shape, (s.shape, rs.shape)  # identical selection (the golden run met 3 near-threshold pairs in all)
        assert np.allclose(s, rs, rtol=1e-3, atol=1e-6)
        assert np.abs(r["box3d_lidar"].cpu().numpy() - rb).max() <= 1e-3
        assert np.array_equal(r["label_preds"].cpu().numpy(), rl)
        assert r["metadata"] == example["metadata"][b]
