"""MultiGroupHead.loss / get_model_ema_loss / consistency_loss of the mirror head on the device (ODIoU and the IoU targets /
teacher-student matching on the HIP kernels) vs tests/golden/head_loss_ref.npz = the REFERENCE's own mg_head_sessd.py,
losses.py, odious.py, box_torch_ops.py, iou3d_utils.py run from source on CPU with only the CUDA extension substituted
(tests/golden/make_golden_head_loss.py). Values 5e-4 relative (ODIoU term 2e-3: float32 reference geometry); gradients of
the total loss w.r.t. the student's head outputs 1e-2 of the largest gradient entry."""
import os

import numpy as np
import pytest
import torch

from sessd_hip import configs

pytestmark = pytest.mark.gpu


def test_head_loss_matches_reference_run(dev, golden_dir):
    from det3d.models import build_detector
    g = np.load(os.path.join(golden_dir, "head_loss_ref.npz"))
    head = build_detector(configs.kitti_car_model(), train_cfg=None, test_cfg=configs.TEST_CFG).bbox_head.to(dev)
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    B = g["labels"].shape[0]
    trans = [dict(flipped=bool(g["trans_flipped"][b]), noise_rotation=float(g["trans_rot"][b]), noise_scale=float(g["trans_scale"][b]))
             for b in range(B)]
    example = dict(anchors=[T("anchors")], anchors_raw=[T("anchors")], labels=[T("labels")], reg_targets=[T("reg_targets")],
                   labels_raw=[T("labels_raw")], reg_targets_raw=[T("reg_targets_raw")], metadata=[{}] * B, transformation=trans)
    stu = {k: T(k + "_stu").clone().requires_grad_(True) for k in ("box", "cls", "dir", "iou")}
    preds = [dict(box_preds=stu["box"], cls_preds=stu["cls"], dir_cls_preds=stu["dir"], iou_preds=stu["iou"])]
    ema = [dict(box_preds=T("box_tea"), cls_preds=T("cls_tea"), dir_cls_preds=T("dir_tea"), iou_preds=T("iou_tea"))]
    ret = head.loss(example, preds, ema)

    def val(k):
        v = ret[k][0]
        return float(v.detach().sum()) if torch.is_tensor(v) else float(v)

    for k, tol in (("loss", 1e-3), ("cls_loss_reduced", 5e-4), ("loc_loss_reduced", 5e-4), ("dir_loss_reduced", 5e-4),
                   ("iou_pred_loss", 5e-4), ("ious_loss", 2e-3), ("cls_pos_loss", 5e-4), ("cls_neg_loss", 5e-4), ("loss_ema", 5e-4),
                   ("cls_loss_reduced_ema", 5e-4), ("iou_pred_loss_ema", 5e-4), ("dir_loss_reduced_ema", 5e-4)):
        want = float(g["ret_" + k])
        assert abs(val(k) - want) <= tol * max(1e-3, abs(want)), (k, val(k), want)
    assert abs(val("consistency_loss") - float(g["ret_consistency_loss"][0])) <= 5e-4 * float(g["ret_consistency_loss"][0])
    assert int(ret["num_pos"][0]) == int(g["ret_num_pos"])
    total = ret["loss"][0] + 1.0 * ret["consistency_loss"][0].sum()  # trainer_sessd.py:267
    total.backward()
    for k in ("box", "cls", "dir", "iou"):
        want = g["grad_" + k]
        got = stu[k].grad.cpu().numpy()
        assert np.abs(got - want).max() <= 1e-2 * np.abs(want).max(), (k, np.abs(got - want).max(), np.abs(want).max())
        assert np.array_equal(got != 0, want != 0) or np.abs(got - want).max() < 1e-6  # same support (positives / matched boxes)
