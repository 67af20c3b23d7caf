"""mirrors the parts of det3d/core/sampler/preprocess.py that the SE-SSD training pipeline executes (config.py:140-165):
ground-truth database sampling support (BatchSampler, the two database filters), the BEV collision test, the per-object
noise with collision-checked retries (noise_per_object_v4_), the global flip / rotation / scaling that also records the
`transformation` the consistency loss undoes, and the ground-truth range filters.

Host stage, as in the reference (numpy in a DataLoader worker there): box-level decisions involve <= a few dozen boxes, the
point-level work is one broadcast per call. Random numbers come from numpy's global generator in the reference's call
order, so the same seed reproduces the reference's augmentation. Pinned by tests/golden/datapath_ref.npz."""
import numpy as np

from det3d.core.bbox import box_np_ops
from det3d.core.bbox.geometry import points_in_convex_polygon_3d_jit, points_in_convex_polygon_jit


class BatchSampler(object):
    """Walks a shuffled index list in chunks; when a request reaches the end it returns the (possibly short) tail and
    reshuffles (preprocess.py:20-58)."""

    def __init__(self, sampled_list, name=None, epoch=None, shuffle=True, drop_reminder=False):
        self._sampled_list = sampled_list
        self._indices = np.arange(len(sampled_list))
        if shuffle:
            np.random.shuffle(self._indices)
        self._idx, self._example_num, self._name, self._shuffle = 0, len(sampled_list), name, shuffle

    def _sample(self, num):
        if self._idx + num >= self._example_num:
            ret = self._indices[self._idx:].copy()
            if self._shuffle:
                np.random.shuffle(self._indices)
            self._idx = 0
        else:
            ret = self._indices[self._idx:self._idx + num]
            self._idx += num
        return ret

    def sample(self, num):
        return [self._sampled_list[i] for i in self._sample(num)]


class DBFilterByDifficulty(object):
    """drops database objects whose difficulty is listed (preprocess.py:71-84)."""

    def __init__(self, removed_difficulties, logger=None):
        self._removed = removed_difficulties

    def __call__(self, db_infos):
        return {k: [i for i in v if i["difficulty"] not in self._removed] for k, v in db_infos.items()}


class DBFilterByMinNumPoint(object):
    """keeps database objects with at least min points, per class (preprocess.py:87-100)."""

    def __init__(self, min_gt_point_dict, logger=None):
        self._min = min_gt_point_dict

    def __call__(self, db_infos):
        for name, min_num in self._min.items():
            if min_num > 0:
                db_infos[name] = [i for i in db_infos[name] if i["num_points_in_gt"] >= min_num]
        return db_infos


class DataBasePreprocessor(object):
    def __init__(self, preprocessors):
        self._preprocessors = preprocessors

    def __call__(self, db_infos):
        for p in self._preprocessors:
            db_infos = p(db_infos)
        return db_infos


def filter_gt_box_outside_range(gt_boxes, limit_range):
    """mask of boxes with at least one BEV corner strictly inside [x0, y0, x1, y1] (preprocess.py:138-148)."""
    bev = box_np_ops.center_to_corner_box2d(gt_boxes[:, [0, 1]], gt_boxes[:, [3, 4]], gt_boxes[:, -1])
    region = box_np_ops.minmax_to_corner_2d(np.asarray(limit_range)[np.newaxis, ...])
    return points_in_convex_polygon_jit(bev.reshape(-1, 2), region).reshape(-1, 4).any(axis=1)


def filter_gt_box_outside_range_by_center(gt_boxes, limit_range):
    """mask of boxes whose centre is strictly inside the range (preprocess.py:167-179)."""
    region = box_np_ops.minmax_to_corner_2d(np.asarray(limit_range)[np.newaxis, ...])
    return points_in_convex_polygon_jit(gt_boxes[:, :2], region).reshape(-1)


def filter_gt_low_points(gt_boxes, points, num_gt_points, point_num_threshold=2):
    """drops boxes with <= threshold points together with those points (preprocess.py:182-191)."""
    weak = np.asarray(num_gt_points) <= point_num_threshold
    if weak.any():
        points = points[~box_np_ops.points_in_rbbox(points, gt_boxes[weak]).any(axis=1)]
    return gt_boxes[~weak], points


def mask_points_in_corners(points, box_corners):
    return points_in_convex_polygon_3d_jit(points[:, :3], box_np_ops.corner_to_surfaces_3d(box_corners))


def _native():
    """libsessd_hip.so's host functions for the box-level decisions (csrc/host_boxes.hip), or None if the library cannot be
    loaded in this process (then the numpy forms below run: same decisions, ~10x the time)."""
    global _NATIVE
    if _NATIVE is None:
        try:
            from sessd_hip._lib import lib
            _NATIVE = lib if hasattr(lib, "sessd_box_collision_host") else False
        except Exception:  # noqa: BLE001 -- a data-loader worker without the library still works
            _NATIVE = False
    return _NATIVE or None


_NATIVE = None
USE_NATIVE_BOX_OPS = True


def box_collision_test(boxes, qboxes, clockwise=True):
    """(N, 4, 2) x (K, 4, 2) BEV quadrilaterals -> (N, K) bool: their bounding rectangles overlap and either two edges cross
    or one quadrilateral lies inside the other (preprocess.py:944-1027). Runs sessd_box_collision_host (plain C++, the same
    arithmetic in the arrays' precision) when both arrays are float32 or both float64; box_collision_test_numpy otherwise."""
    lib = _native() if USE_NATIVE_BOX_OPS else None
    if lib is not None and boxes.dtype == qboxes.dtype and boxes.dtype in (np.float32, np.float64) and boxes.ndim == 3 \
            and boxes.shape[1:] == (4, 2) and qboxes.shape[1:] == (4, 2):
        a, b = np.ascontiguousarray(boxes), np.ascontiguousarray(qboxes)
        out = np.empty((a.shape[0], b.shape[0]), dtype=np.uint8)
        rc = lib.sessd_box_collision_host(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], 1 if a.dtype == np.float32 else 0,
                                          1 if clockwise else 0, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("sessd_box_collision_host: %d" % rc)
        return out.astype(np.bool_)
    return box_collision_test_numpy(boxes, qboxes, clockwise)


def box_collision_test_numpy(boxes, qboxes, clockwise=True):
    """The vectorised numpy form of box_collision_test (the branch structure of the reference's numba kernel -- compiled,
    `flag is False` compares values -- collapses to this expression)."""
    sa, sb = box_np_ops.corner_to_standup_nd_jit(boxes), box_np_ops.corner_to_standup_nd_jit(qboxes)
    iw = np.minimum(sa[:, None, 2], sb[None, :, 2]) - np.maximum(sa[:, None, 0], sb[None, :, 0])
    ih = np.minimum(sa[:, None, 3], sb[None, :, 3]) - np.maximum(sa[:, None, 1], sb[None, :, 1])
    near = (iw > 0) & (ih > 0)
    A, B = boxes[:, None, :, None, :], np.roll(boxes, -1, axis=1)[:, None, :, None, :]       # edge k of box i
    C, D = qboxes[None, :, None, :, :], np.roll(qboxes, -1, axis=1)[None, :, None, :, :]    # edge l of box j

    def ccw(P, Q, R):
        return (R[..., 1] - P[..., 1]) * (Q[..., 0] - P[..., 0]) > (Q[..., 1] - P[..., 1]) * (R[..., 0] - P[..., 0])

    crossing = ((ccw(A, C, D) != ccw(B, C, D)) & (ccw(A, B, C) != ccw(A, B, D))).any(axis=(2, 3))

    def inside(outer, inner):   # [o, i]: every corner of inner[i] strictly inside outer[o]
        vec = outer - np.roll(outer, -1, axis=1)
        if clockwise:
            vec = -vec
        cross = (vec[:, None, :, None, 1] * (outer[:, None, :, None, 0] - inner[None, :, None, :, 0])
                 - vec[:, None, :, None, 0] * (outer[:, None, :, None, 1] - inner[None, :, None, :, 1]))
        return ~(cross >= 0).any(axis=(2, 3))

    return near & (crossing | inside(boxes, qboxes) | inside(qboxes, boxes).T)


def noise_per_box(boxes, valid_mask, loc_noises, rot_noises):
    """(N, 5) BEV boxes, candidate noises (N, T, 3) / (N, T) -> per box the index of the first candidate whose moved,
    rotated footprint collides with no other box in its current place, -1 if none does. Boxes are processed in order and an
    accepted move is what later boxes are tested against; boxes outside valid_mask stay but still block (preprocess.py:579-611)."""
    n = boxes.shape[0]
    corners = box_np_ops.box2d_to_corner_jit(boxes)
    lib = _native() if USE_NATIVE_BOX_OPS else None
    if lib is not None and n and boxes.dtype in (np.float32, np.float64) and rot_noises.shape[1]:
        # the sequential loop in C++ (sessd_noise_per_box_host); sines / cosines of the candidates from numpy, so that the
        # candidate footprints are the numpy form's bit for bit
        corners = np.ascontiguousarray(corners)
        centers = np.ascontiguousarray(boxes[:, :2])
        valid = np.ascontiguousarray(np.asarray(valid_mask, dtype=np.uint8))
        rot = np.ascontiguousarray(rot_noises, dtype=np.float64)
        loc_xy = np.ascontiguousarray(np.asarray(loc_noises, dtype=np.float64)[:, :, :2])
        sin_r, cos_r = np.sin(rot), np.cos(rot)
        chosen = np.empty((n,), dtype=np.int64)
        rc = lib.sessd_noise_per_box_host(corners.ctypes.data, centers.ctypes.data, valid.ctypes.data, loc_xy.ctypes.data,
                                          sin_r.ctypes.data, cos_r.ctypes.data, n, rot.shape[1], 1 if boxes.dtype == np.float32 else 0,
                                          chosen.ctypes.data)
        if rc != 0:
            raise RuntimeError("sessd_noise_per_box_host: %d" % rc)
        return chosen
    chosen = -np.ones((n,), dtype=np.int64)
    for i in range(n):
        if not valid_mask[i]:
            continue
        local = corners[i] - boxes[i, :2]
        t0, step, T = 0, 4, rot_noises.shape[1]
        while t0 < T and chosen[i] < 0:   # candidates in growing chunks: the first few almost always contain a free one
            t1 = min(T, t0 + step)
            s, c = np.sin(rot_noises[i, t0:t1])[:, None], np.cos(rot_noises[i, t0:t1])[:, None]
            cand = np.empty((t1 - t0, 4, 2), dtype=boxes.dtype)
            cand[..., 0] = local[None, :, 0] * c + local[None, :, 1] * s
            cand[..., 1] = local[None, :, 0] * -s + local[None, :, 1] * c
            cand += (boxes[i, :2] + loc_noises[i, t0:t1, :2])[:, None, :]
            hit = box_collision_test_numpy(cand, corners)
            hit[:, i] = False
            free = np.nonzero(~hit.any(axis=1))[0]
            if free.size:
                chosen[i] = t0 + free[0]
                corners[i] = cand[free[0]]
            t0, step = t1, step * 4
    return chosen


def _select_transform(transform, indices):
    """row i = transform[i, indices[i]], zeros where indices[i] is -1 (preprocess.py:571-576)."""
    out = np.zeros((transform.shape[0],) + transform.shape[2:], dtype=transform.dtype)
    ok = indices != -1
    out[ok] = transform[np.nonzero(ok)[0], indices[ok]]
    return out


def points_transform_(points, centers, point_masks, loc_transform, rot_transform, valid_mask):
    """in place: every point takes the motion of the FIRST valid box that contains it -- rotation about that box's centre,
    then the box's translation (preprocess.py:544-560)."""
    m = point_masks.astype(bool) & np.asarray(valid_mask, bool)[None, :]
    moved = m.any(axis=1)
    if not moved.any():
        return
    j = m[moved].argmax(axis=1)
    dt = points.dtype   # the reference updates the float32 rows in place: every step is rounded back to the point dtype
    s, c = np.sin(rot_transform[j]).astype(dt), np.cos(rot_transform[j]).astype(dt)
    p = (points[moved, :3] - centers[j, :3]).astype(dt)
    q = np.empty_like(p)
    q[:, 0] = p[:, 0] * c + p[:, 1] * s
    q[:, 1] = p[:, 0] * -s + p[:, 1] * c
    q[:, 2] = p[:, 2]
    q = (q + centers[j, :3]).astype(dt)
    q = (q + loc_transform[j]).astype(dt)
    points[moved, :3] = q


def box3d_transform_(boxes, loc_transform, rot_transform, valid_mask):
    """in place: translation and yaw change of the valid boxes (preprocess.py:562-568)."""
    v = np.asarray(valid_mask, bool)
    boxes[v, :3] += loc_transform[v]
    boxes[v, 6] += rot_transform[v]


def noise_per_object_v4_(gt_boxes, points=None, valid_mask=None, rotation_perturb=np.pi / 4, center_noise_std=1.0,
                         global_random_rot_range=np.pi / 4, num_try=5, group_ids=None, data_aug_with_context=-1.0,
                         data_aug_random_drop=-1.0):
    """in place: every valid ground-truth box and the points inside it get an independent random shift
    (normal, std center_noise_std) and yaw change (uniform in rotation_perturb), the first of num_try draws that keeps the
    footprint free of collisions (preprocess.py:615-658)."""
    n = gt_boxes.shape[0]
    std = np.array(center_noise_std, dtype=gt_boxes.dtype)
    loc_noises = np.random.normal(scale=std, size=[n, num_try, 3])
    rot_noises = np.random.uniform(rotation_perturb[0], rotation_perturb[1], size=[n, num_try])
    ctx = data_aug_with_context if data_aug_with_context > 0 else 0.0
    offset = [0.0, 0.0, ctx, ctx, 0.0]
    corners = box_np_ops.center_to_corner_box3d(gt_boxes[:, :3], gt_boxes[:, 3:6] + offset[2:5], gt_boxes[:, 6],
                                                origin=[0.5, 0.5, 0.5], axis=2)
    chosen = noise_per_box(gt_boxes[:, [0, 1, 3, 4, 6]] + offset, valid_mask, loc_noises, rot_noises)
    loc_t, rot_t = _select_transform(loc_noises, chosen), _select_transform(rot_noises, chosen)
    surfaces = box_np_ops.corner_to_surfaces_3d_jit(corners)
    if callable(points):
        # device mode of the pipeline stage: the caller moves the points (sessd_points_rigid_moves) given the faces of the boxes
        # BEFORE the move, their centres and the chosen transforms -- same random draws, same box updates
        points(surfaces, gt_boxes[:, :3].copy(), loc_t, rot_t, valid_mask)
    elif points is not None:
        masks = points_in_convex_polygon_3d_jit(points[:, :3], surfaces)
        points_transform_(points, gt_boxes[:, :3], masks, loc_t, rot_t, valid_mask)
    box3d_transform_(gt_boxes, loc_t, rot_t, valid_mask)


def random_flip_v2(gt_boxes, points, probability=0.5):
    """mirror about the x axis (y -> -y, yaw -> pi - yaw) with the given probability; returns the decision (preprocess.py:896-905)."""
    enable = np.random.choice([False, True], replace=False, p=[1 - probability, probability])
    if enable:
        if gt_boxes is not None:
            gt_boxes[:, 1] = -gt_boxes[:, 1]
            gt_boxes[:, -1] = -gt_boxes[:, -1] + np.pi
            if gt_boxes.shape[1] > 7:
                gt_boxes[:, 7] = -gt_boxes[:, 7]
        points[:, 1] = -points[:, 1]
    return gt_boxes, points, enable


def global_rotation_v3(gt_boxes, points, rotation=np.pi / 4):
    """one yaw rotation of the whole scene, uniform in the range; returns the angle (preprocess.py:930-941)."""
    if not isinstance(rotation, list):
        rotation = [-rotation, rotation]
    angle = np.random.uniform(rotation[0], rotation[1])
    points[:, :3] = box_np_ops.rotation_points_single_angle(points[:, :3], angle, axis=2)
    if gt_boxes is not None:
        gt_boxes[:, :3] = box_np_ops.rotation_points_single_angle(gt_boxes[:, :3], angle, axis=2)
        if gt_boxes.shape[1] > 7:
            v = np.hstack([gt_boxes[:, 6:8], np.zeros((gt_boxes.shape[0], 1))])
            gt_boxes[:, 6:8] = box_np_ops.rotation_points_single_angle(v, angle, axis=2)[:, :2]
        gt_boxes[:, -1] += angle
    return gt_boxes, points, angle


def global_scaling_v3(gt_boxes, points, min_scale=0.95, max_scale=1.05):
    """one uniform scale of coordinates and box sizes; returns the factor (preprocess.py:914-919)."""
    scale = np.random.uniform(min_scale, max_scale)
    points[:, :3] *= scale
    if gt_boxes is not None:
        gt_boxes[:, :-1] *= scale
    return gt_boxes, points, scale
