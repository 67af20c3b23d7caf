"""Point-in-convex-body tests of det3d/core/bbox/geometry.py that the data pipeline uses (GT-AUG point removal, per-object noise,
shape-aware augmentation, ground-truth range filter). The reference runs them as numba loops with early exits; here each is one
broadcast over (points, bodies, faces) with the same elementwise arithmetic, so the masks are identical.
Pinned by tests/golden/datapath_ref.npz (the reference functions run from source)."""
import numpy as np


def surface_equ_3d_jitv2(surfaces):
    """(B, F, >=3, 3) face vertices -> inward-pointing normals (B, F, 3) and offsets d (B, F) of n.x + d = 0
    (geometry.py:352-377): n = (v0 - v1) x (v1 - v2), d = -v0 . n."""
    e0 = surfaces[:, :, 0] - surfaces[:, :, 1]
    e1 = surfaces[:, :, 1] - surfaces[:, :, 2]
    n = np.empty(surfaces.shape[:2] + (3,), dtype=surfaces.dtype)
    n[..., 0] = e0[..., 1] * e1[..., 2] - e0[..., 2] * e1[..., 1]
    n[..., 1] = e0[..., 2] * e1[..., 0] - e0[..., 0] * e1[..., 2]
    n[..., 2] = e0[..., 0] * e1[..., 1] - e0[..., 1] * e1[..., 0]
    v0 = surfaces[:, :, 0]
    d = -v0[..., 0] * n[..., 0] - v0[..., 1] * n[..., 1] - v0[..., 2] * n[..., 2]
    return n, d


_CULL_PAD = 1e-2   # metres; far above the float32 rounding of the plane test, so culled points are outside by that test too


def points_in_convex_polygon_3d_jit(points, polygon_surfaces, num_surfaces=None):
    """(P, 3) points, (B, F, V, 3) faces with inward normals -> (P, B) bool: strictly inside every face, n.p + d < 0
    (geometry.py:215-276). `num_surfaces` (faces actually used per body) cuts the face list like the reference's loop bound.
    The plane test (same arithmetic as the reference's loop) only runs on the points inside each body's padded bounding box,
    found through one sort of the points along x: a frame has ~20 k points and <= ~100 bodies a few metres across."""
    P, B = points.shape[0], polygon_surfaces.shape[0]
    out = np.zeros((P, B), dtype=np.bool_)
    if P == 0 or B == 0:
        return out
    n, d = surface_equ_3d_jitv2(polygon_surfaces[:, :, :3, :])
    used = None
    if num_surfaces is not None:   # face k is looked at while k <= num_surfaces[body]
        used = np.arange(polygon_surfaces.shape[1])[None, :] <= np.asarray(num_surfaces)[:, None]
    verts = polygon_surfaces.reshape(B, -1, 3)
    lo, hi = verts.min(axis=1) - _CULL_PAD, verts.max(axis=1) + _CULL_PAD
    order = np.argsort(points[:, 0])
    xs = points[order, 0]
    first, last = np.searchsorted(xs, lo[:, 0], side="left"), np.searchsorted(xs, hi[:, 0], side="right")
    for b in range(B):
        cand = order[first[b]:last[b]]
        if cand.size == 0:
            continue
        q = points[cand]
        cand = cand[(q[:, 1] >= lo[b, 1]) & (q[:, 1] <= hi[b, 1]) & (q[:, 2] >= lo[b, 2]) & (q[:, 2] <= hi[b, 2])]
        if cand.size == 0:
            continue
        q = points[cand]
        sign = q[:, None, 0] * n[None, b, :, 0] + q[:, None, 1] * n[None, b, :, 1] + q[:, None, 2] * n[None, b, :, 2] + d[None, b]
        outside = sign >= 0
        if used is not None:
            outside = outside & used[None, b]
        out[cand, b] = ~outside.any(axis=1)
    return out


def points_in_convex_polygon_jit(points, polygon, clockwise=True):
    """(P, 2) points, (B, V, 2) convex polygons -> (P, B) bool: strictly on the inner side of every edge (geometry.py:279-325)."""
    prev = np.roll(polygon, 1, axis=1)
    vec = polygon - prev if clockwise else prev - polygon
    cross = (vec[None, :, :, 1] * (polygon[None, :, :, 0] - points[:, None, None, 0])
             - vec[None, :, :, 0] * (polygon[None, :, :, 1] - points[:, None, None, 1]))
    return ~(cross >= 0).any(axis=2)
