"""Anchor grid and camera-frustum helpers of the predict path (host side, numpy; built once per model).

Mirrors det3d/core/bbox/box_np_ops.py:780-833 (create_anchors_3d_range) and :995-1004 (get_valid_frustum
with projection_matrix_to_CRT_kitti :620-634, get_frustum :637-654, camera_to_lidar, corner_to_surfaces_3d_jit)."""
import numpy as np


def create_anchors_3d_range(feature_size=(1, 200, 176), anchor_range=(0, -40.0, -1.0, 70.4, 40.0, -1.0),
                            sizes=(1.6, 3.9, 1.56), rotations=(0, 1.57), dtype=np.float32):
    """(D,H,W,num_sizes,num_rot,7) [x,y,z,w,l,h,r]. NB: the x stride offsets BOTH x and y centres (as the reference)."""
    ar = np.array(anchor_range, dtype)
    D, H, W = [int(v) for v in feature_size]
    stride = (ar[3] - ar[0]) / W
    zc = np.linspace(ar[2], ar[5], D, dtype=dtype)
    yc = np.linspace(ar[1], ar[4], H, endpoint=False, dtype=dtype) + stride / 2
    xc = np.linspace(ar[0], ar[3], W, endpoint=False, dtype=dtype) + stride / 2
    sizes = np.reshape(np.array(sizes, dtype), [-1, 3])
    rot = np.array(rotations, dtype)
    out = np.zeros((D, H, W, sizes.shape[0], rot.shape[0], 7), dtype)
    out[..., 0] = xc.reshape(1, 1, W, 1, 1)
    out[..., 1] = yc.reshape(1, H, 1, 1, 1)
    out[..., 2] = zc.reshape(D, 1, 1, 1, 1)
    out[..., 3:6] = sizes.reshape(1, 1, 1, -1, 1, 3)
    out[..., 6] = rot.reshape(1, 1, 1, 1, -1)
    return out


def projection_matrix_to_CRT_kitti(proj):
    CR, CT = proj[0:3, 0:3], proj[0:3, 3]
    Rinv, Cinv = np.linalg.qr(np.linalg.inv(CR))
    return np.linalg.inv(Cinv), np.linalg.inv(Rinv), Cinv @ CT


def get_valid_frustum(rect, Trv2c, P2, image_shape, near_clip=0.001, far_clip=100):
    """(1,6,4,3) float64: the 6 inward-facing quads of the camera frustum in lidar coordinates."""
    C, R, T = projection_matrix_to_CRT_kitti(P2)
    b = [0, 0, image_shape[1], image_shape[0]]
    fku, fkv, u0v0 = C[0, 0], -C[1, 1], C[0:2, 2]
    z = np.array([near_clip] * 4 + [far_clip] * 4, dtype=C.dtype)[:, None]
    bc = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[3]], [b[2], b[1]]], dtype=C.dtype)
    nb = (bc - u0v0) / np.array([fku / near_clip, -fkv / near_clip], dtype=C.dtype)
    fb = (bc - u0v0) / np.array([fku / far_clip, -fkv / far_clip], dtype=C.dtype)
    fr = np.concatenate([np.concatenate([nb, fb], 0), z], 1) - T
    pts = (np.linalg.inv(R) @ fr.T).T
    pts = np.concatenate([pts, np.ones((pts.shape[0], 1))], -1)
    corners = (pts @ np.linalg.inv((rect @ Trv2c).T))[..., :3]
    idx = np.array([0, 1, 2, 3, 7, 6, 5, 4, 0, 3, 7, 4, 1, 5, 6, 2, 0, 4, 5, 1, 3, 2, 6, 7]).reshape(6, 4)
    return corners[idx][None]
