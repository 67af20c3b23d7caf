"""Deterministic synthetic KITTI-shaped inputs: frames, weights, calib.

There is no KITTI data and no released checkpoint in this environment, so the benchmark and the
parity tests run on ray-cast point clouds of the KITTI front-camera field of view and on seeded
random weights. numpy only (frame_labels: the numpy box helpers of the det3d mirror); nothing here touches the device.
"""
import math

import numpy as np

KITTI_RANGE = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]
KITTI_VOXEL = [0.05, 0.05, 0.1]


def _ray_boxes(dirs, centers, dims, yaws):
    """Nearest hit distance of rays from the origin against rotated boxes. dirs (R,3) unit vectors."""
    R = dirs.shape[0]
    best = np.full(R, np.inf)
    for c, d, yaw in zip(centers, dims, yaws):
        co, si = math.cos(yaw), math.sin(yaw)
        rot = np.array([[co, si, 0], [-si, co, 0], [0, 0, 1.0]])
        o = rot @ (-np.asarray(c))
        dl = dirs @ rot.T
        half = np.asarray(d) * 0.5
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (-half - o) / dl
            t2 = (half - o) / dl
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tmax >= tmin) & (tmax > 0) & (tmin > 0)
        best = np.where(hit & (tmin < best), tmin, best)
    return best


def make_frame(seed=0, num_points=20000, supersample=1):
    """HDL-64E-like front-FOV scan: 64 beams (+2 .. -24.8 deg), azimuth +-45 deg at 0.1755 deg,
    ground plane z=-1.73 m, 6 vertical walls, 15 car-sized boxes, 1 cm range noise, U(0,1) intensity.
    Returns (P,4) float32 [x,y,z,r] in scan order (azimuth-major), subsampled to num_points."""
    rng = np.random.RandomState(seed)
    nb = 64 * supersample
    elev = np.deg2rad(np.linspace(2.0, -24.8, nb))
    naz = int(round(90.0 / 0.1755)) * supersample + 1
    az = np.deg2rad(np.linspace(-45.0, 45.0, naz))
    A, E = np.meshgrid(az, elev, indexing="ij")  # azimuth-major like a spinning sensor
    dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    t = np.full(dirs.shape[0], np.inf)
    # ground
    with np.errstate(divide="ignore"):
        tg = np.where(dirs[:, 2] < 0, -1.73 / dirs[:, 2], np.inf)
    t = np.minimum(t, tg)
    # walls: thin tall boxes
    wc, wd, wy = [], [], []
    for _ in range(6):
        r = rng.uniform(15, 65)
        a = rng.uniform(-0.7, 0.7)
        wc.append([r * math.cos(a), r * math.sin(a), -0.2])
        wd.append([rng.uniform(6, 25), 0.3, 3.0])
        wy.append(rng.uniform(-math.pi, math.pi))
    t = np.minimum(t, _ray_boxes(dirs, wc, wd, wy))
    # cars
    cc, cd, cy = [], [], []
    for _ in range(15):
        cc.append([rng.uniform(5, 60), rng.uniform(-25, 25), -1.73 + 0.78])
        cd.append([3.9, 1.6, 1.56])
        cy.append(rng.uniform(-math.pi, math.pi))
    t = np.minimum(t, _ray_boxes(dirs, cc, cd, cy))
    ok = np.isfinite(t) & (t < 80.0)
    t = t[ok] + rng.normal(0, 0.01, ok.sum())
    pts = dirs[ok] * t[:, None]
    inten = rng.uniform(0, 1, pts.shape[0])
    out = np.concatenate([pts, inten[:, None]], 1).astype(np.float32)
    if num_points is not None and out.shape[0] > num_points:
        idx = np.sort(rng.choice(out.shape[0], num_points, replace=False))
        out = out[idx]
    return np.ascontiguousarray(out)


def frame_cars(seed=0):
    """The 15 car-sized boxes make_frame(seed) places in its scene, as KITTI lidar boxes (15, 7) [x, y, z, w, l, h, r] (the same
    random draws in the same order as make_frame: six walls first). The ray caster turns a car's LONG side (3.9 m) to
    (cos yaw, sin yaw); a det3d lidar box turns its length axis to (sin r, cos r) (box_np_ops.rotation_3d_in_axis, axis 2), so
    r = pi/2 - yaw, folded into [-pi, pi). (Rounds 1 - 4 returned r = yaw: boxes a quarter turn off the points they label -- it
    did not matter to a timing benchmark on random weights, it does to training.) Ground truth of the synthetic training set."""
    rng = np.random.RandomState(seed)
    for _ in range(6):
        rng.uniform(15, 65); rng.uniform(-0.7, 0.7); rng.uniform(6, 25); rng.uniform(-math.pi, math.pi)
    out = np.zeros((15, 7), np.float32)
    for i in range(15):
        x, y = rng.uniform(5, 60), rng.uniform(-25, 25)
        r = math.pi / 2 - rng.uniform(-math.pi, math.pi)
        out[i] = [x, y, -1.73 + 0.78, 1.6, 3.9, 1.56, (r + math.pi) % (2 * math.pi) - math.pi]
    return out


def frame_labels(seed, points, margin=0.1):
    """(boxes (15, 7), point counts (15,)) of make_frame(seed)'s cars for the cloud `points` (the subsampled scan): how many of
    its points lie inside each car's box grown by `margin` (the rays end ON the surfaces, with 1 cm of range noise). A car behind
    a wall or outside the +-45 degree field of view has none: what a labeller could not see is not ground truth."""
    from det3d.core.bbox import box_np_ops
    cars = frame_cars(seed)
    grown = cars.copy()
    grown[:, 3:6] += margin
    counts = box_np_ops.points_in_rbbox(points[:, :3], grown, z_axis=2, origin=(0.5, 0.5, 0.5)).sum(0)
    return cars, counts.astype(np.int64)


def kitti_calib():
    """A KITTI-typical calibration (P2 fx=fy=721.5377, cx=609.5593, cy=172.854; image 375x1242)."""
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884],
                   [0, 0, 0, 1]], np.float64)
    rect = np.array([[0.9999239, 0.00983776, -0.007445048, 0], [-0.009869795, 0.9999421, -0.004278459, 0],
                     [0.007402527, 0.004351614, 0.9999631, 0], [0, 0, 0, 1]], np.float64)
    Trv2c = np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766],
                      [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
                      [0.9998621, 0.00752379, 0.01480755, -0.2717806], [0, 0, 0, 1]], np.float64)
    return dict(P2=P2, rect=rect, Trv2c=Trv2c, image_shape=(375, 1242))


def random_boxes7(n, seed=0):
    """(n,7) [x,y,z,w,l,h,r] car-like boxes for the IoU / NMS micro cases."""
    rng = np.random.RandomState(seed)
    x = rng.uniform(0, 70, n)
    y = rng.uniform(-40, 40, n)
    z = rng.normal(-1.0, 0.3, n)
    w = np.abs(rng.normal(1.6, 0.2, n)) + 0.2
    l = np.abs(rng.normal(3.9, 0.2, n)) + 0.2
    h = np.abs(rng.normal(1.56, 0.2, n)) + 0.2
    r = rng.uniform(-math.pi, math.pi, n)
    return np.stack([x, y, z, w, l, h, r], 1).astype(np.float32)


def clustered_boxes7(n, seed=0, clusters=None):
    """Boxes bunched around a few centres so that many pairs really overlap (NMS-shaped input)."""
    rng = np.random.RandomState(seed)
    k = clusters or max(1, n // 12)
    cx = rng.uniform(5, 65, k)
    cy = rng.uniform(-35, 35, k)
    cr = rng.uniform(-math.pi, math.pi, k)
    a = rng.randint(0, k, n)
    b = random_boxes7(n, seed + 1)
    b[:, 0] = cx[a] + rng.normal(0, 0.5, n)
    b[:, 1] = cy[a] + rng.normal(0, 0.5, n)
    b[:, 6] = cr[a] + rng.normal(0, 0.15, n)
    return b.astype(np.float32)


def boxes7_to_bev5(b):
    """[x,y,z,w,l,h,r] -> [x-w/2, y-l/2, x+w/2, y+l/2, r]  (det3d/core/iou3d/utils.py:74-101, 'wlh')."""
    b = np.asarray(b, np.float32)
    return np.stack([b[:, 0] - b[:, 3] / 2, b[:, 1] - b[:, 4] / 2, b[:, 0] + b[:, 3] / 2, b[:, 1] + b[:, 4] / 2,
                     b[:, 6]], 1).astype(np.float32)


def boxes7_to_bev7(b):
    """[x,y,z,w,l,h,r] -> [x1,y1,z1,x2,y2,z2,r] with z +- h/2 (iou3d_utils.py:173-178)."""
    b = np.asarray(b, np.float32)
    return np.stack([b[:, 0] - b[:, 3] / 2, b[:, 1] - b[:, 4] / 2, b[:, 2] - b[:, 5] / 2, b[:, 0] + b[:, 3] / 2,
                     b[:, 1] + b[:, 4] / 2, b[:, 2] + b[:, 5] / 2, b[:, 6]], 1).astype(np.float32)


def init_synthetic_weights(model, seed=0):
    """Deterministic non-trivial weights for a det3d-mirror VoxelNet (no checkpoint exists here): kaiming-uniform
    convolutions, BatchNorm gamma~U(0.5,1.5), beta~U(-0.1,0.1); small box/dir/iou head weights. BatchNorm running
    statistics and the classification bias are then set by `calibrate_synthetic_model` on one frame, which is what
    keeps activations O(1) through the 28 layers (as in a trained network)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                if name.startswith("backbone"):
                    fan_in = p.shape[-2] * int(np.prod(p.shape[:-2]))
                elif "deconv" in name:
                    fan_in = p.shape[0] * p.shape[2] * p.shape[3] / 4.0
                else:
                    fan_in = p.shape[1] * int(np.prod(p.shape[2:]))
                bound = math.sqrt(6.0 / fan_in)
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
        for name, m in model.named_modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                c = m.num_features
                m.weight.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.copy_(torch.rand(c, generator=g) * 0.2 - 0.1)
                m.running_mean.zero_()
                m.running_var.fill_(1.0)
        head = model.bbox_head.tasks[0]
        head.conv_box.weight.mul_(0.3)
        head.conv_box.bias.zero_()
        head.conv_dir.bias.zero_()
        head.conv_iou.bias.fill_(0.5)
        head.conv_iou.weight.mul_(0.3)
        head.conv_cls.bias.zero_()
    model.eval()
    return model


def calibrate_synthetic_model(model, voxel_features, coors, batch_size, input_shape, pass_fraction=0.007,
                              sparse_runner=None):
    """Set every BatchNorm's running mean/var to the statistics of its input on the given frame(s), layer by
    layer (data-dependent init), then choose the classification bias so that `pass_fraction` of the anchors
    clear the 0.3 score threshold. Runs on whatever device the model lives on, through the nn modules themselves
    (spconv-shim sparse convs on the HIP device, torch convs for the dense neck). Test / benchmark set-up only."""
    import torch
    import torch.nn.functional as F
    hooks = []

    def pre(mod, inp):
        x = inp[0].detach().float()
        dims = [d for d in range(x.dim()) if d != 1]
        mod.running_mean.copy_(x.mean(dims))
        mod.running_var.copy_(x.var(dims, unbiased=False).clamp_min(1e-6))

    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            hooks.append(m.register_forward_pre_hook(pre))
    try:
        with torch.no_grad():
            run = sparse_runner if sparse_runner is not None else model.backbone
            x = run(voxel_features, coors, batch_size, input_shape)
            nk = model.neck
            x0 = nk.bottom_up_block_0(x)
            x1 = nk.bottom_up_block_1(x0)
            t0, t1 = nk.trans_0(x0), nk.trans_1(x1)
            m0 = nk.deconv_block_0(t1) + t0
            m1 = nk.deconv_block_1(t1)
            o0, o1 = nk.conv_0(m0), nk.conv_1(m1)
            w = torch.softmax(torch.cat([nk.w_0(o0), nk.w_1(o1)], 1), 1)
            out = o0 * w[:, 0:1] + o1 * w[:, 1:]
            head = model.bbox_head.tasks[0]
            head.conv_cls.bias.zero_()
            logits = F.conv2d(out, head.conv_cls.weight).reshape(-1)
            q = torch.quantile(logits.float().cpu(), 1.0 - pass_fraction)
            head.conv_cls.bias.fill_(float(math.log(0.3 / 0.7) - q))
            # box codes ~N(0, 0.1) and IoU predictions ~N(0.5, 0.15): plausible, well-conditioned boxes
            for conv, target, bias in ((head.conv_box, 0.1, 0.0), (head.conv_iou, 0.15, 0.5), (head.conv_dir, 1.0, 0.0)):
                y = F.conv2d(out, conv.weight)
                conv.weight.mul_(target / float(y.std().clamp_min(1e-6)))
                conv.bias.fill_(bias)
    finally:
        for h in hooks:
            h.remove()
    model.eval()
    return model
