"""Deterministic synthetic KITTI-shaped inputs: frames, weights, calib.

There is no KITTI data and no released checkpoint in this environment, so the benchmark and the
parity tests run on ray-cast point clouds of the KITTI front-camera field of view and on seeded
random weights. numpy only; nothing here touches the device.
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
