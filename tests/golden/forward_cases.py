"""Seeded inputs shared by tests/golden/make_golden_forward.py (which runs the REFERENCE's sources on them, build container
only) and the tests (which run the oracle / the mirror modules / the kernels on the SAME inputs). Large tensors (weights of the
SSFA neck: 2.9 M floats; head outputs over 70400 anchors) are regenerated from their seed instead of being stored; the golden
file keeps check values of them, so a drift of this generator is caught rather than silently compared against itself."""
import numpy as np
import torch


def seeded_state_dict(shapes, seed):
    """shapes: {key: shape} (reference state_dict names). Deterministic values by sorted key: conv weights uniform with a
    fan-in scale, BatchNorm weight in [0.5,1.5], bias / running_mean small, running_var in [0.5,1.5], counters 0."""
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shp, dtype=torch.long)
        elif k.endswith("running_var"):
            out[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith("running_mean"):
            out[k] = torch.randn(shp, generator=g) * 0.1
        elif len(shp) == 1 and k.endswith("weight"):  # BatchNorm gamma
            out[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith("bias"):
            out[k] = torch.randn(shp, generator=g) * 0.1
        else:  # conv / deconv weight
            fan = int(np.prod(shp[1:])) if len(shp) > 1 else 1
            out[k] = (torch.rand(shp, generator=g) * 2 - 1) * float(np.sqrt(3.0 / max(fan, 1))) * 1.4
    return out


def ssfa_input(seed=3, B=2, H=16, W=12):
    g = torch.Generator().manual_seed(int(seed))
    x = torch.randn(B, 128, H, W, generator=g)
    return torch.relu(x) * (torch.rand(B, 1, H, W, generator=g) > 0.4)  # sparse-ish non-negative BEV map, like .dense() output


def head_input(seed=4, B=2, H=8, W=6):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(B, 128, H, W, generator=g)


def vfe_case(seed=5, M=700, T=5):
    rng = np.random.RandomState(seed)
    num = rng.randint(1, T + 1, size=M).astype(np.int32)
    vox = (rng.randn(M, T, 4) * np.array([20, 20, 1, 0.3]) + np.array([35, 0, -1, 0.5])).astype(np.float32)
    for i in range(M):
        vox[i, num[i]:] = 0  # zero padding of the unused slots, as the voxelizer leaves it
    return vox, num


def predict_case(seed, B=2, H=200, W=176):
    """Head outputs of B frames over the full KITTI car anchor grid (2 anchors per cell, MultiGroupHead hard-codes 70400):
    NHWC tensors as Head.forward returns them. A few dozen 'objects' raise the class logits of the anchors around them, so that
    several hundred anchors pass the 0.3 score threshold in overlapping clusters (what rotated NMS at IoU 0.01 has to resolve)."""
    rng = np.random.RandomState(seed)
    A = 2
    box = (rng.randn(B, H, W, A * 7) * 0.15).astype(np.float32)
    cls = (rng.randn(B, H, W, A) * 0.8 - 6.0).astype(np.float32)
    dirp = rng.randn(B, H, W, A * 2).astype(np.float32)
    iou = rng.uniform(-0.6, 1.0, (B, H, W, A)).astype(np.float32)
    for b in range(B):
        for _ in range(28 + 6 * b):
            cy, cx = rng.randint(4, H - 4), rng.randint(4, W - 4)
            r = rng.randint(1, 4)
            ys, xs = slice(cy - r, cy + r + 1), slice(cx - r, cx + r + 1)
            cls[b, ys, xs, :] += rng.uniform(4.0, 9.0, (2 * r + 1, 2 * r + 1, A)).astype(np.float32)
        # boxes near the limits of post_center_range / far corners, and exact ties of the score
        cls[b, 0, 0, :] = 5.0
        cls[b, H - 1, W - 1, :] = 5.0
        box[b, 0, 0, 2] = -40.0   # z residual pushes the centre below the range -> dropped by the range filter
        cls[b, 100, 50:54, 0] = 3.25
        iou[b, 100, 50:54, 0] = 0.5
    return dict(box_preds=box, cls_preds=cls, dir_cls_preds=dirp, iou_preds=iou)


def collate_samples(seed=7, n_samples=2):
    """Per-sample `res` dicts as they reach Reformat in validation and in labelled training mode (preprocess.py:196-232,
    :236-358 produce them): small voxel sets, anchors / targets of one task, calib, annotations, raw twins + transformation."""
    rng = np.random.RandomState(seed)
    out = []
    for s in range(n_samples):
        def vox(m):
            return dict(voxels=rng.randn(m, 5, 4).astype(np.float32), coordinates=rng.randint(0, 40, (m, 3)).astype(np.int32),
                        num_points=rng.randint(1, 6, m).astype(np.int32), num_voxels=np.array([m], dtype=np.int64),
                        shape=np.array([1408, 1600, 40], dtype=np.int64))
        m, mr, na = 50 + 7 * s, 44 + 5 * s, 64
        targets = dict(anchors=[rng.randn(na, 7).astype(np.float32)], labels=[rng.randint(-1, 2, na).astype(np.int32)],
                       reg_targets=[rng.randn(na, 7).astype(np.float32)], reg_weights=[rng.rand(na).astype(np.float32)],
                       positive_gt_id=[rng.randint(-1, 5, na).astype(np.int32)])
        targets_raw = dict(anchors=[targets["anchors"][0].copy()], labels=[rng.randint(-1, 2, na).astype(np.int32)],
                           reg_targets=[rng.randn(na, 7).astype(np.float32)], reg_weights=[rng.rand(na).astype(np.float32)],
                           positive_gt_id=[rng.randint(-1, 5, na).astype(np.int32)])
        calib = dict(rect=np.eye(4) + rng.randn(4, 4) * 1e-3, Trv2c=rng.randn(4, 4), P2=rng.randn(4, 4),
                     frustum=rng.randn(1, 6, 4, 3))
        lidar = dict(points=rng.randn(120 + 11 * s, 4).astype(np.float32), voxels=vox(m), targets=targets,
                     annotations=dict(gt_boxes=[rng.randn(3 + s, 7).astype(np.float32)], gt_names=[np.array(["Car"] * (3 + s))]),
                     points_raw=rng.randn(110 + 9 * s, 4).astype(np.float32), voxels_raw=vox(mr), targets_raw=targets_raw,
                     annotations_raw=dict(gt_boxes=[rng.randn(3 + s, 7).astype(np.float32)], gt_names=[np.array(["Car"] * (3 + s))]),
                     transformation=dict(flipped=bool(s % 2), noise_rotation=float(rng.uniform(-0.3, 0.3)),
                                         noise_scale=float(rng.uniform(0.95, 1.05))))
        out.append(dict(lidar=lidar, calib=calib,
                        metadata=dict(image_prefix="/data", num_point_features=4, image_idx=100 + s, image_shape=np.array([375, 1242]),
                                      token=str(100 + s))))
    return out
