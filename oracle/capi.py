"""TEST INFRASTRUCTURE ONLY. ctypes front-end of oracle/liboracle.so and oracle/_ref."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_ref = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        from . import build
        _lib = C.CDLL(build.build_oracle())
        L = _lib
        L.oracle_points_to_voxel.restype = C.c_int
        L.oracle_points_to_voxel.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, _f32p, _i32p, _i32p]
        L.oracle_vfe_mean.argtypes = [_f32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.oracle_rect_overlap.restype = C.c_float
        L.oracle_rect_overlap.argtypes = [C.c_float] * 10
        for n in ("oracle_boxes_overlap_bev", "oracle_boxes_iou_bev"):
            getattr(L, n).argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p]
        L.oracle_boxes_aligned_overlap_bev.argtypes = [_f32p, _f32p, C.c_int, _f32p]
        L.oracle_boxes_iou3d.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
        L.oracle_nms_sorted.restype = C.c_int
        L.oracle_nms_sorted.argtypes = [_f32p, C.c_int, C.c_float, C.c_int, _i64p]
        L.oracle_box2d_corners.argtypes = [_f32p, _f32p]
        L.oracle_quad_iou.restype = C.c_double
        L.oracle_quad_iou.argtypes = [_f32p, _f32p]
        L.oracle_rotate_nms.restype = C.c_int
        L.oracle_rotate_nms.argtypes = [_f32p, C.c_int, _i32p, C.c_float, _i32p, C.c_double, _i32p]
        L.oracle_rotate_nms_pairs.argtypes = [_f32p, C.c_int, _i32p, C.c_float, _i32p, C.c_double, _i32p, _i32p, C.c_int]
        L.oracle_rotate_nms_forced.argtypes = [_f32p, C.c_int, _i32p, C.c_float, _i32p, C.c_double, _i32p, _i32p, C.c_int, _i32p, C.c_int]
        L.oracle_rotate_iou_eval.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int, _f32p]
        L.oracle_rotate_iou_pair.restype = C.c_float
        L.oracle_rotate_iou_pair.argtypes = [_f32p, _f32p, C.c_int]
        L.oracle_di_nms.restype = C.c_int
        L.oracle_di_nms.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _i32p, _i32p, C.c_void_p, C.c_int, C.c_float, _f32p, C.c_int,
                                    _f32p, C.c_float, C.c_int, _f32p, _f32p, _i32p, _i32p, _i32p]
    return _lib


def ref_lib():
    """The compiled REFERENCE iou3d CPU path (None when it was never built)."""
    global _ref
    if _ref is None:
        from . import build
        p = build.build_ref()
        if p is None or not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
        for n in ("ref_boxes_overlap_bev_cpu", "ref_boxes_iou_bev_cpu", "ref_boxes_iou3d_cpu"):
            f = getattr(_ref, n)
            f.restype = C.c_int
            f.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p]
    return _ref


_ref_nms = None


def ref_nms_module():
    """The reference's det3d/ops/nms/nms_cpu.h compiled from source with the boost::geometry shim (oracle/build.py
    build_ref_nms): module with rotate_non_max_suppression_cpu / IOU_weighted_rotate_non_max_suppression_cpu, or None."""
    global _ref_nms
    if _ref_nms is None:
        from . import build
        p = build.build_ref_nms()
        if p is None or not os.path.exists(p):
            return None
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_nms_cpu", p)
        _ref_nms = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_ref_nms)
    return _ref_nms


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- voxelizer
def points_to_voxel(points, voxel_size, coors_range, max_points=35, max_voxels=20000):
    """point_cloud_ops_v2.py:120-194 (reverse_index=True). Returns voxels, coors[z,y,x], num_points."""
    points = _f32(points)
    N, ndim = points.shape
    vs, cr = _f32(voxel_size), _f32(coors_range)
    voxels = np.empty((max_voxels, max_points, ndim), np.float32)
    coors = np.empty((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    m = lib().oracle_points_to_voxel(points, N, ndim, vs, cr, max_points, max_voxels, voxels, coors, num)
    if m < 0:
        raise MemoryError("oracle voxel map")
    return voxels[:m], coors[:m], num[:m]


def vfe_mean(voxels, num_points, nfeat=4):
    voxels = _f32(voxels)
    M, MP, ndim = voxels.shape
    out = np.empty((M, nfeat), np.float32)
    lib().oracle_vfe_mean(voxels, np.ascontiguousarray(num_points, np.int32), M, MP, ndim, nfeat, out)
    return out


# ---------------------------------------------------------------- iou3d
def _pair(fn, a, b, wa):
    a, b = _f32(a).reshape(-1, wa), _f32(b).reshape(-1, wa)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    fn(a, a.shape[0], b, b.shape[0], out)
    return out


def boxes_overlap_bev(a, b):
    return _pair(lib().oracle_boxes_overlap_bev, a, b, 5)


def boxes_iou_bev(a, b):
    return _pair(lib().oracle_boxes_iou_bev, a, b, 5)


def boxes_aligned_overlap_bev(a, b):
    a, b = _f32(a).reshape(-1, 5), _f32(b).reshape(-1, 5)
    out = np.zeros((a.shape[0],), np.float32)
    lib().oracle_boxes_aligned_overlap_bev(a, b, a.shape[0], out)
    return out


def boxes_iou3d(a, b, gpu_variant=True):
    a, b = _f32(a).reshape(-1, 7), _f32(b).reshape(-1, 7)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().oracle_boxes_iou3d(a, a.shape[0], b, b.shape[0], out, 1 if gpu_variant else 0)
    return out


def nms_sorted(boxes, thresh, mode):
    """mode 0 rotated bev (N,5), 1 3-D (N,7), 2 axis-aligned (N,5); boxes sorted by score desc."""
    boxes = _f32(boxes)
    keep = np.zeros((boxes.shape[0],), np.int64)
    n = lib().oracle_nms_sorted(boxes, boxes.shape[0], float(thresh), mode, keep)
    return keep[:n]


def ref_boxes_overlap_bev(a, b):
    return _pair(ref_lib().ref_boxes_overlap_bev_cpu, a, b, 5)


def ref_boxes_iou_bev(a, b):
    return _pair(ref_lib().ref_boxes_iou_bev_cpu, a, b, 5)


def ref_boxes_iou3d(a, b):
    return _pair(ref_lib().ref_boxes_iou3d_cpu, a, b, 7)


# ---------------------------------------------------------------- rotated NMS of predict()
def box2d_corners(dets):
    dets = _f32(dets)
    out = np.zeros((dets.shape[0], 4, 2), np.float32)
    for i in range(dets.shape[0]):
        row = np.ascontiguousarray(dets[i, :5])
        lib().oracle_box2d_corners(row, out[i].reshape(-1))
    return out


def quad_iou(p, q):
    return lib().oracle_quad_iou(_f32(p).reshape(-1), _f32(q).reshape(-1))


def rotate_nms_cc(dets, thresh, order=None, margin=1e-4, return_pairs=False, forced=None):
    """nms_cpu.py:40-51. dets (K,6) [x,y,w,l,r,score]. Returns (keep, n_near_threshold_pairs) and, with return_pairs, the
    (n,2) array of (kept box, candidate) indices whose IoU lies within `margin` of the threshold. forced: (m,3) rows
    (i, j, suppress) imposing the decision of pair (kept i, candidate j) (oracle/compare.py explores the marginal ones)."""
    dets = _f32(dets)
    K = dets.shape[0]
    if order is None:
        # descending score, ties by ascending index (the reference leaves ties unspecified)
        order = np.lexsort((np.arange(K), -dets[:, 5].astype(np.float64))).astype(np.int32)
    order = np.ascontiguousarray(order, np.int32)
    keep = np.zeros((max(K, 1),), np.int32)
    near = np.zeros((1,), np.int32)
    if return_pairs or forced is not None:
        pairs = np.zeros((4096, 2), np.int32)
        fz = np.ascontiguousarray(np.asarray(forced if forced is not None and len(forced) else np.zeros((0, 3)), np.int32).reshape(-1, 3))
        n = lib().oracle_rotate_nms_forced(dets, K, order, float(thresh), keep, float(margin), near, pairs.reshape(-1), 4096,
                                           fz.reshape(-1) if len(fz) else np.zeros((3,), np.int32), int(len(fz)))
        return keep[:n].astype(np.int64), int(near[0]), pairs[:min(int(near[0]), 4096)].astype(np.int64)
    n = lib().oracle_rotate_nms(dets, K, order, float(thresh), keep, float(margin), near)
    return keep[:n].astype(np.int64), int(near[0])


def di_nms_core(boxes, box_corners, standup_iou, thresh, scores, iou_preds, labels, dirs, anchors, cnt_thresh,
                nms_sigma_dist_interval, nms_sigma_square, suppressed_thresh, centerness_c):
    """Stand-in for the pybind IOU_weighted_rotate_non_max_suppression_cpu (nms_cpu.h:173-384), same argument list and the same
    [boxes, scores, labels, dirs, keep] list-of-lists return (`thresh` is unused by the reference as well)."""
    b = _f32(boxes).reshape(-1, 7)
    n = b.shape[0]
    co = _f32(box_corners).reshape(n, 8)
    su = _f32(standup_iou).reshape(n, n)
    an = _f32(anchors) if int(centerness_c) == 1 else None
    if an is not None:
        an = np.ascontiguousarray(an.reshape(n, -1))
    iv, sg = _f32(np.asarray(nms_sigma_dist_interval, np.float32)), _f32(np.asarray(nms_sigma_square, np.float32))
    bo, so = np.zeros((max(n, 1), 7), np.float32), np.zeros((max(n, 1),), np.float32)
    lo, do, ke = np.zeros((max(n, 1),), np.int32), np.zeros((max(n, 1),), np.int32), np.zeros((max(n, 1),), np.int32)
    k = lib().oracle_di_nms(b, co, su, n, _f32(scores), _f32(iou_preds), np.ascontiguousarray(labels, np.int32),
                            np.ascontiguousarray(dirs, np.int32), None if an is None else an.ctypes.data_as(C.c_void_p),
                            0 if an is None else an.shape[1], float(cnt_thresh), iv, int(iv.shape[0]), sg, float(suppressed_thresh),
                            int(centerness_c), bo, so, lo, do, ke)
    return [bo[:k].tolist(), so[:k].tolist(), lo[:k].tolist(), do[:k].tolist(), ke[:k].tolist()]


# ---------------------------------------------------------------- numba-convention rotated IoU (AP evaluation family)
def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """nms_gpu.py:636-672 rotate_iou_gpu_eval: boxes (N,5), query (K,5) [cx,cy,w,l,angle] -> (N,K)."""
    b, q = _f32(boxes).reshape(-1, 5), _f32(query_boxes).reshape(-1, 5)
    out = np.zeros((b.shape[0], q.shape[0]), np.float32)
    lib().oracle_rotate_iou_eval(b, b.shape[0], q, q.shape[0], int(criterion), out)
    return out


def rotate_iou_pair(r1, r2, criterion=-1):
    return float(lib().oracle_rotate_iou_pair(_f32(r1), _f32(r2), int(criterion)))
