"""The det3d-mirror boundary (CPU): config loading, registry building, reference state_dict layout."""
import os

import pytest
import torch

REF_CFG = "/root/reference/examples/second/configs/config.py"


def _build():
    from det3d.models import build_detector
    from sessd_hip import configs
    return build_detector(configs.kitti_car_model(), train_cfg=None, test_cfg=configs.TEST_CFG)


def test_reference_config_loads_unchanged():
    if not os.path.exists(REF_CFG):
        pytest.skip("reference tree absent (GPU box)")
    from det3d.torchie import Config
    from det3d.models import build_detector
    cfg = Config.fromfile(REF_CFG)
    assert cfg.model.type == "VoxelNet" and cfg.test_cfg.nms.nms_iou_threshold == 0.01
    assert cfg.model.bbox_head.box_coder.n_dim == 7          # a real object survives inside the config tree
    assert cfg.get("nonexistent", 5) == 5
    with pytest.raises(AttributeError):
        cfg.model.nonexistent
    assert "voxel_generator" in cfg.text and cfg.filename == REF_CFG
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert sum(p.numel() for p in m.parameters()) == 3811674  # SURVEY.md section 2b / BASELINE.md
    # the in-repo dict config is the same model
    sd_a = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd_b = {k: tuple(v.shape) for k, v in _build().state_dict().items()}
    assert sd_a == sd_b


def test_state_dict_layout_matches_reference_checkpoint_keys():
    sd = _build().state_dict()
    assert tuple(sd["backbone.middle_conv.0.weight"].shape) == (3, 3, 3, 4, 16)       # spconv [kz,ky,kx,Cin,Cout]
    assert tuple(sd["backbone.middle_conv.39.weight"].shape) == (3, 1, 1, 64, 64)
    for i in range(14):
        assert "backbone.middle_conv.%d.weight" % (3 * i) in sd
        assert "backbone.middle_conv.%d.running_var" % (3 * i + 1) in sd
    for k in ("neck.bottom_up_block_0.1.weight", "neck.bottom_up_block_0.2.running_mean", "neck.bottom_up_block_0.7.weight",
              "neck.bottom_up_block_1.0.weight", "neck.bottom_up_block_1.6.weight", "neck.trans_0.0.weight",
              "neck.trans_1.1.bias", "neck.deconv_block_0.0.weight", "neck.deconv_block_1.1.weight", "neck.conv_0.0.weight",
              "neck.w_0.0.weight", "neck.w_1.1.running_var", "bbox_head.tasks.0.conv_box.weight",
              "bbox_head.tasks.0.conv_cls.bias", "bbox_head.tasks.0.conv_dir.weight", "bbox_head.tasks.0.conv_iou.bias"):
        assert k in sd, k
    assert tuple(sd["neck.deconv_block_0.0.weight"].shape) == (256, 128, 3, 3)
    assert tuple(sd["bbox_head.tasks.0.conv_box.weight"].shape) == (14, 128, 1, 1)
    assert sum(v.numel() for k, v in sd.items() if "num_batches" not in k and "running" not in k) == 3811674


def test_registry_and_builders():
    from det3d.models.registry import BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, READERS
    for reg, name in ((DETECTORS, "VoxelNet"), (READERS, "VoxelFeatureExtractorV3"), (BACKBONES, "SpMiddleFHD"),
                      (NECKS, "SSFA"), (NECKS, "RPN"), (HEADS, "MultiGroupHead"), (LOSSES, "SigmoidFocalLoss"),
                      (LOSSES, "WeightedSmoothL1Loss"), (LOSSES, "WeightedSoftmaxClassificationLoss")):
        assert reg.get(name) is not None
    from det3d.utils import build_from_cfg
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="NoSuchNeck"), NECKS)
    from det3d.utils.config_tool import get_downsample_factor
    from sessd_hip import configs
    assert get_downsample_factor(configs.kitti_car_model()) == 8
    from det3d.core.input.voxel_generator import VoxelGenerator
    vg = VoxelGenerator([0.05, 0.05, 0.1], [0, -40.0, -3.0, 70.4, 40.0, 1.0], 5, 20000)
    assert list(vg.grid_size) == [1408, 1600, 40]


def test_box_coder_roundtrip():
    from det3d.builder import build_box_coder
    bc = build_box_coder(dict(type="ground_box3d_coder", n_dim=7, linear_dim=False, encode_angle_vector=False))
    g = torch.Generator().manual_seed(0)
    anchors = torch.rand(50, 7, generator=g) + 0.5
    boxes = torch.rand(50, 7, generator=g) + 0.5
    assert torch.allclose(bc.decode_torch(bc.encode_torch(boxes, anchors), anchors), boxes, atol=1e-5)


def test_box_np_ops_helpers_match_reference_numpy(golden_dir):
    """det3d.core.bbox.box_np_ops (mirror) vs the reference's own functions run from source (nms_helpers_ref.npz)."""
    import os
    import numpy as np
    from det3d.core.bbox import box_np_ops
    g = np.load(os.path.join(golden_dir, "nms_helpers_ref.npz"))
    d = g["dets"]
    c = box_np_ops.center_to_corner_box2d(d[:, :2], d[:, 2:4], d[:, 4])
    assert c.shape == (64, 4, 2) and np.allclose(c, g["corners"], atol=2e-6, rtol=0)
    su = box_np_ops.corner_to_standup_nd(c)
    assert np.allclose(su, g["standup"], atol=2e-6, rtol=0)
    iou = box_np_ops.iou_jit(g["standup"], g["standup"], eps=0.0)
    assert np.allclose(iou, g["standup_iou"], atol=1e-6, rtol=1e-5)
    assert np.array_equal(iou > 0, g["standup_iou"] > 0)


def test_loss_modules_match_reference_run(golden_dir):
    """det3d.models.losses (mirror) vs the reference's losses.py run from source (losses_ref.npz): values and gradients."""
    import os
    import numpy as np
    import torch
    from det3d.models import losses as L
    g = np.load(os.path.join(golden_dir, "losses_ref.npz"))
    T = lambda k: torch.from_numpy(g[k])

    x = T("focal_logits").requires_grad_(True)
    y = L.SigmoidFocalLoss(gamma=2.0, alpha=0.25)(x, T("focal_targets"), weights=T("focal_w"))
    y.sum().backward()
    assert np.allclose(y.detach().numpy(), g["focal_out"], rtol=1e-5, atol=1e-7)
    assert np.allclose(x.grad.numpy(), g["focal_grad"], rtol=1e-4, atol=1e-7)

    p = T("sl1_pred").requires_grad_(True)
    y = L.WeightedSmoothL1Loss(sigma=3.0, code_weights=[1.0] * 7, codewise=True, loss_weight=2.0)(p, T("sl1_tgt"), weights=T("focal_w"))
    y.sum().backward()
    assert np.allclose(y.detach().numpy(), g["sl1_out"], rtol=1e-5, atol=1e-7)
    assert np.allclose(p.grad.numpy(), g["sl1_grad"], rtol=1e-5, atol=1e-7)

    d = T("dir_logits").requires_grad_(True)
    y = L.WeightedSoftmaxClassificationLoss(name="direction_classifier", loss_weight=0.2)(d, T("dir_tgt"), weights=T("focal_w"))
    y.sum().backward()
    assert np.allclose(y.detach().numpy(), g["dir_out"], rtol=1e-5, atol=1e-7)
    assert np.allclose(d.grad.numpy(), g["dir_grad"], rtol=1e-5, atol=1e-7)


def test_second_box_codec_matches_reference_run(golden_dir):
    """box_torch_ops.second_box_decode / second_box_encode (mirror) vs decode_ref.npz (the reference's function run from source)."""
    import os
    import numpy as np
    import torch
    from det3d.core.bbox import box_torch_ops as B
    g = np.load(os.path.join(golden_dir, "decode_ref.npz"))
    dec = B.second_box_decode(torch.from_numpy(g["enc"]), torch.from_numpy(g["anchors"]))
    assert np.allclose(dec.numpy(), g["dec"], rtol=1e-6, atol=1e-6)
    enc = B.second_box_encode(dec, torch.from_numpy(g["anchors"]))
    assert np.allclose(enc.numpy(), g["enc"], rtol=1e-5, atol=2e-6)
