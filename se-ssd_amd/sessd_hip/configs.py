"""The SE-SSD KITTI-car model / test / voxel settings as plain dicts (values of examples/second/configs/config.py:
model :48-90, test_cfg :113-124, voxel_generator :174-180), for tests and the benchmark on machines where the
reference tree (and therefore its config file) is absent. `Config.fromfile(<reference config.py>)` gives the same."""
import logging

from det3d.builder import build_box_coder


def kitti_car_model():
    tasks = [dict(num_class=1, class_names=["Car"])]
    box_coder = dict(type="ground_box3d_coder", n_dim=7, linear_dim=False, encode_angle_vector=False)
    return dict(
        type="VoxelNet", pretrained=None,
        reader=dict(type="VoxelFeatureExtractorV3", num_input_features=4, norm_cfg=None),
        backbone=dict(type="SpMiddleFHD", num_input_features=4, ds_factor=8, norm_cfg=None),
        neck=dict(type="SSFA", layer_nums=[5], ds_layer_strides=[1], ds_num_filters=[128], us_layer_strides=[1],
                  us_num_filters=[128], num_input_features=128, norm_cfg=None, logger=logging.getLogger("RPN")),
        bbox_head=dict(type="MultiGroupHead", mode="3d", in_channels=128, norm_cfg=None, tasks=tasks, weights=[1],
                       box_coder=build_box_coder(box_coder), encode_background_as_zeros=True,
                       loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=1.0),
                       loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0),
                       use_sigmoid_score=True,
                       loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=[1.0] * 7, codewise=True, loss_weight=2.0),
                       encode_rad_error_by_sin=True,
                       loss_aux=dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier", loss_weight=0.2),
                       direction_offset=0.0))


TEST_CFG = dict(nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000, nms_post_max_size=100,
                         nms_iou_threshold=0.01),
                score_threshold=0.3, post_center_limit_range=[0, -40.0, -5.0, 70.4, 40.0, 5.0], max_per_img=100)

VOXEL_GENERATOR = dict(range=[0, -40.0, -3.0, 70.4, 40.0, 1.0], voxel_size=[0.05, 0.05, 0.1], max_points_in_voxel=5,
                       max_voxel_num=20000)


def build_synthetic_detector(device, seed=0, calib_frame_seed=0, max_voxels=16000, num_points=20000, supersample=1):
    """det3d-mirror VoxelNet with seeded weights, BatchNorm statistics calibrated on one synthetic frame (on `device`) of the
    workload's own density (supersample = 3 for the 200 k-point dense scenes): random weights calibrated on a sparse scan give
    activations (and decoded boxes) of absurd magnitude on a dense one."""
    import torch
    from det3d.models import build_detector
    from . import ops, synth
    model = build_detector(kitti_car_model(), train_cfg=None, test_cfg=TEST_CFG)
    synth.init_synthetic_weights(model, seed)
    model.to(device)
    pts = torch.from_numpy(synth.make_frame(calib_frame_seed, num_points, supersample=supersample)).to(device)
    r = ops.voxelize_batch([pts], VOXEL_GENERATOR["voxel_size"], VOXEL_GENERATOR["range"], 5, max_voxels)
    m = int(r["prefix"][1].item())
    synth.calibrate_synthetic_model(model, r["mean"][:m].contiguous(), r["coors"][:m].contiguous(), 1, [1408, 1600, 40])
    return model
