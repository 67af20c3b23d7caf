"""numpy box helpers of det3d/core/bbox/box_np_ops.py that the SE-SSD inference path and its NMS wrappers touch
(anchors / frustum live in sessd_hip.anchors). Vectorised restatements; pinned by tests/golden/nms_helpers_ref.npz."""
import numpy as np

from sessd_hip.anchors import create_anchors_3d_range, get_valid_frustum, projection_matrix_to_CRT_kitti  # noqa: F401


def corners_nd(dims, origin=0.5):
    """(N, ndim) sizes -> (N, 2**ndim, ndim) corner offsets around `origin` (box_np_ops.py:433-463); 2-D order:
    (0,0), (0,1), (1,1), (1,0) in units of the box size."""
    dims = np.asarray(dims)
    ndim = dims.shape[1]
    unit = np.stack(np.unravel_index(np.arange(2 ** ndim), [2] * ndim), axis=1).astype(dims.dtype)
    if ndim == 2:
        unit = unit[[0, 1, 3, 2]]
    elif ndim == 3:
        unit = unit[[0, 1, 3, 2, 4, 5, 7, 6]]
    unit = unit - np.asarray(origin, dtype=dims.dtype)
    return dims[:, None, :] * unit[None, :, :]


def rotation_2d(points, angles):
    """(N, P, 2) points rotated by angles (N,), clockwise-positive convention of box_np_ops.py:267-294:
    x' = x cos a + y sin a, y' = -x sin a + y cos a."""
    s, c = np.sin(angles)[:, None], np.cos(angles)[:, None]
    x, y = points[..., 0], points[..., 1]
    return np.stack([x * c + y * s, -x * s + y * c], axis=-1)


def center_to_corner_box2d(centers, dims, angles=None, origin=0.5):
    """box_np_ops.py:512-532"""
    corners = corners_nd(dims, origin=origin)
    if angles is not None:
        corners = rotation_2d(corners, angles)
    return corners + np.asarray(centers).reshape(-1, 1, 2)


def corner_to_standup_nd(boxes_corner):
    """(N, P, ndim) corners -> (N, 2*ndim) [mins, maxs] (box_np_ops.py:346-351)."""
    assert boxes_corner.ndim == 3
    return np.concatenate([boxes_corner.min(axis=1), boxes_corner.max(axis=1)], axis=-1)


def iou_jit(boxes, query_boxes, eps=1.0):
    """Axis-aligned IoU matrix (N, K) with eps-widened extents; zero where the widened overlap is not positive in both
    axes (box_np_ops.py:1008-1046)."""
    b, q = np.asarray(boxes), np.asarray(query_boxes)
    iw = np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0]) + eps
    ih = np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1]) + eps
    area_b = (b[:, 2] - b[:, 0] + eps) * (b[:, 3] - b[:, 1] + eps)
    area_q = (q[:, 2] - q[:, 0] + eps) * (q[:, 3] - q[:, 1] + eps)
    ok = (iw > 0) & (ih > 0)
    inter = np.where(ok, iw * ih, 0).astype(b.dtype)
    ua = area_b[:, None] + area_q[None, :] - inter
    return np.where(ok, inter / np.where(ok, ua, 1), 0).astype(b.dtype)


def _corners_and_standup_iou(rbboxes, qrbboxes):
    boxes_corners = center_to_corner_box2d(rbboxes[:, :2], rbboxes[:, 2:4], rbboxes[:, 4])
    qboxes_corners = center_to_corner_box2d(qrbboxes[:, :2], qrbboxes[:, 2:4], qrbboxes[:, 4])
    # if the stand-up boxes do not overlap, the rotated boxes do not either
    standup_iou = iou_jit(corner_to_standup_nd(boxes_corners), corner_to_standup_nd(qboxes_corners), eps=0.0)
    return boxes_corners, qboxes_corners, standup_iou


def riou_cc(rbboxes, qrbboxes, standup_thresh=0.0):
    """rotated BEV IoU of (N,5) x (K,5) [x,y,w,l,r] boxes (box_np_ops.py:20-32) through spconv.utils.rbbox_iou"""
    from spconv.utils import rbbox_iou
    return rbbox_iou(*_corners_and_standup_iou(rbboxes, qrbboxes), standup_thresh)


def rinter_cc(rbboxes, qrbboxes, standup_thresh=0.0):
    """rotated BEV intersection area (box_np_ops.py:35-50) through spconv.utils.rbbox_intersection"""
    from spconv.utils import rbbox_intersection
    return rbbox_intersection(*_corners_and_standup_iou(rbboxes, qrbboxes), standup_thresh)


def limit_period(val, offset=0.5, period=2 * np.pi):
    """val folded into [-offset*period, (1-offset)*period) (box_np_ops.py:619-620)."""
    return val - np.floor(val / period + offset) * period


def lidar_to_camera(points, r_rect, velo2cam):
    """(…,3) lidar points -> rectified camera frame: [p, 1] (R0_rect Tr_velo_to_cam)^T (box_np_ops.py:945-965)."""
    if points.shape[-1] == 3:
        points = np.concatenate([points, np.ones(list(points.shape[:-1]) + [1])], axis=-1)
    return (points @ (r_rect @ velo2cam).T)[..., :3]


def box_lidar_to_camera(data, r_rect, velo2cam):
    """[x,y,z,w,l,h,r] lidar -> [x',y',z',l,h,w,r] camera (box_np_ops.py:973-978)."""
    xyz = lidar_to_camera(data[:, 0:3], r_rect, velo2cam)
    return np.concatenate([xyz, data[:, 4:5], data[:, 5:6], data[:, 3:4], data[:, 6:7]], axis=1)


def rotation_3d_in_axis(points, angles, axis=0):
    """(N,P,3) points rotated per box about `axis` with the reference's (clockwise-positive) matrices (box_np_ops.py:369-405)."""
    s, c = np.sin(angles), np.cos(angles)
    one, zero = np.ones_like(c), np.zeros_like(c)
    if axis == 1:
        m = [[c, zero, -s], [zero, one, zero], [s, zero, c]]
    elif axis in (2, -1):
        m = [[c, -s, zero], [s, c, zero], [zero, zero, one]]
    elif axis == 0:
        m = [[zero, c, -s], [zero, s, c], [one, zero, zero]]
    else:
        raise ValueError("axis should in range")
    return np.einsum("aij,jka->aik", points, np.stack(m))


def center_to_corner_box3d(centers, dims, angles=None, origin=(0.5, 0.5, 0.5), axis=2):
    """(N,3) centres, (N,3) sizes, yaw -> (N,8,3) corners; origin = where the centre sits inside the box
    ([0.5, 1.0, 0.5] camera, [0.5, 0.5, 0] lidar), axis = rotation axis (1 camera, 2 lidar) (box_np_ops.py:467-503)."""
    corners = corners_nd(dims, origin=origin)
    if angles is not None:
        corners = rotation_3d_in_axis(corners, angles, axis=axis)
    return corners + np.asarray(centers).reshape(-1, 1, 3)


def project_to_image(points_3d, proj_mat):
    """(…,3) camera points -> pixels with the 4x4 P2: [p, 0] P^T, divided by depth (box_np_ops.py:928-934; the homogeneous
    coordinate appended by the reference is 0, not 1: the translation column of P2 is ignored)."""
    p4 = np.concatenate([points_3d, np.zeros(list(points_3d.shape[:-1]) + [1])], axis=-1)
    uvw = p4 @ proj_mat.T
    return uvw[..., :2] / uvw[..., 2:3]


# ---- helpers of the training data path ---------------------------------------------------------------------
_FACE_CORNERS = np.array([0, 1, 2, 3, 7, 6, 5, 4, 0, 3, 7, 4, 1, 5, 6, 2, 0, 4, 5, 1, 3, 2, 6, 7]).reshape(6, 4)


def corner_to_surfaces_3d(corners):
    """(N, 8, 3) box corners of center_to_corner_box3d -> (N, 6, 4, 3) faces whose normals point inwards
    (box_np_ops.py:1160-1189 and its jit twin :1192-1212)."""
    return corners[:, _FACE_CORNERS]


corner_to_surfaces_3d_jit = corner_to_surfaces_3d


def points_in_rbbox(points, rbbox, z_axis=2, origin=(0.5, 0.5, 0.5)):
    """(P, >=3) points, (N, 7) [x,y,z,w,l,h,r] boxes -> (P, N) bool membership (box_np_ops.py:1152-1157)."""
    from det3d.core.bbox.geometry import points_in_convex_polygon_3d_jit
    corners = center_to_corner_box3d(rbbox[:, :3], rbbox[:, 3:6], rbbox[:, -1], origin=origin, axis=z_axis)
    return points_in_convex_polygon_3d_jit(points[:, :3], corner_to_surfaces_3d(corners))


def points_count_rbbox(points, rbbox, z_axis=2, origin=(0.5, 0.5, 0.5)):
    """points per box (box_np_ops.py:12-17)."""
    return points_in_rbbox(points, rbbox, z_axis, origin).sum(axis=0)


def box2d_to_corner_jit(boxes):
    """(N, 5) [x, y, w, l, r] -> (N, 4, 2) BEV corners, order (-,-), (-,+), (+,+), (+,-) in box axes, rotated with
    x' = x cos r + y sin r, y' = -x sin r + y cos r (box_np_ops.py:535-566)."""
    unit = np.array([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]], dtype=boxes.dtype)
    local = boxes[:, None, 2:4] * unit[None]
    s, c = np.sin(boxes[:, -1])[:, None], np.cos(boxes[:, -1])[:, None]
    out = np.empty_like(local)
    out[..., 0] = local[..., 0] * c + local[..., 1] * s
    out[..., 1] = local[..., 0] * -s + local[..., 1] * c
    return out + boxes[:, None, :2]


def corner_to_standup_nd_jit(boxes_corner):
    """(N, P, ndim) -> (N, 2 ndim) [mins, maxs] (box_np_ops.py:319-329)."""
    return np.concatenate([boxes_corner.min(axis=1), boxes_corner.max(axis=1)], axis=-1)


def minmax_to_corner_2d(minmax_box):
    """(..., 4) [x0, y0, x1, y1] -> corners (box_np_ops.py:581-585)."""
    nd = minmax_box.shape[-1] // 2
    lo = minmax_box[..., :nd]
    return center_to_corner_box2d(lo, minmax_box[..., nd:] - lo, origin=0.0)


def rotation_points_single_angle(points, angle, axis=0):
    """(N, 3) points times the reference's transposed rotation matrix of one angle (box_np_ops.py:408-430)."""
    s, c = np.sin(angle), np.cos(angle)
    if axis == 1:
        m = [[c, 0, -s], [0, 1, 0], [s, 0, c]]
    elif axis in (2, -1):
        m = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    elif axis == 0:
        m = [[1, 0, 0], [0, c, -s], [0, s, c]]
    else:
        raise ValueError("axis should in range")
    return points @ np.array(m, dtype=points.dtype)


def camera_to_lidar(points, r_rect, velo2cam):
    """(…,3) rectified-camera points -> lidar frame (box_np_ops.py:937-942)."""
    if points.shape[-1] == 3:
        points = np.concatenate([points, np.ones(list(points.shape[:-1]) + [1])], axis=-1)
    return (points @ np.linalg.inv((r_rect @ velo2cam).T))[..., :3]


def box_camera_to_lidar(data, r_rect, velo2cam):
    """[x,y,z,l,h,w,r] camera -> [x',y',z',w,l,h,r] lidar (box_np_ops.py:965-970)."""
    xyz = camera_to_lidar(data[:, 0:3], r_rect, velo2cam)
    return np.concatenate([xyz, data[:, 5:6], data[:, 3:4], data[:, 4:5], data[:, 6:7]], axis=1)


def change_box3d_center_(box3d, src, dst):
    """in place: move the reference point of the boxes from `src` to `dst` (fractions of the box size) (box_np_ops.py:1406-1409)."""
    box3d[..., :3] += box3d[..., 3:6] * (np.array(dst, dtype=box3d.dtype) - np.array(src, dtype=box3d.dtype))


def remove_outside_points(points, rect, Trv2c, P2, image_shape):
    """the points whose projection falls inside the camera image: inside the image frustum (near 0.001 m, far 100 m) taken to
    the lidar frame (box_np_ops.py:981-992; what `velodyne_reduced` is made with)."""
    from det3d.core.bbox.geometry import points_in_convex_polygon_3d_jit
    inside = points_in_convex_polygon_3d_jit(points[:, :3], get_valid_frustum(rect, Trv2c, P2, image_shape))
    return points[inside.reshape([-1])]
