"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the SE-SSD orientation-aware distance-IoU loss.

Restates det3d/models/losses/odious.py:837-900 (odiou_3D.forward) and the pieces it composes
    rbbox_to_corners            :448-487   corner order / rotation convention
    compute_vertex              :15-136    corners of one box inside the other + edge/edge intersections (at most 8 points)
    sort_vertex                 :278-318   descending angle around the centroid
    area_polygon                :345-365   fan of |triangle| areas from the first sorted vertex
    mbr_diag_compute            :629-643   scipy ConvexHull of the 8 corners, then mbr_diag_convex_hull :596-626: the
                                           minimum-area rectangle aligned with one of the hull's edges (open chain: the
                                           closing edge is not a candidate; pi is written 3.1415926), its diagonal
in float64 with forward-mode derivatives (value + d/d[x,y,z,w,l,h,r] of the PREDICTED box), so that one pass yields the
per-pair term and the gradient the reference obtains from its hand-written backward functions (:138-276, :320-342,
:367-445) plus autograd. Pinned by tests/golden/odiou_ref.npz = the reference's own code run from source
(tests/golden/make_golden_odiou.py)."""
import math

import numpy as np

PI_REF = 3.1415926  # the literal the reference uses (odious.py:555, 602)
ND = 7


class Dual:
    """value + gradient with respect to the 7 parameters of the predicted box"""
    __slots__ = ("v", "d")

    def __init__(self, v, d=None):
        self.v = float(v)
        self.d = np.zeros(ND) if d is None else d

    @staticmethod
    def var(v, i):
        d = np.zeros(ND)
        d[i] = 1.0
        return Dual(v, d)

    def __add__(self, o):
        return Dual(self.v + o.v, self.d + o.d) if isinstance(o, Dual) else Dual(self.v + o, self.d)
    __radd__ = __add__

    def __neg__(self):
        return Dual(-self.v, -self.d)

    def __sub__(self, o):
        return Dual(self.v - o.v, self.d - o.d) if isinstance(o, Dual) else Dual(self.v - o, self.d)

    def __rsub__(self, o):
        return Dual(o - self.v, -self.d)

    def __mul__(self, o):
        return Dual(self.v * o.v, self.d * o.v + o.d * self.v) if isinstance(o, Dual) else Dual(self.v * o, self.d * o)
    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, Dual):
            return Dual(self.v / o.v, (self.d * o.v - o.d * self.v) / (o.v * o.v))
        return Dual(self.v / o, self.d / o)

    def __rtruediv__(self, o):
        return Dual(o / self.v, -o * self.d / (self.v * self.v))


def _c(x):
    return x if isinstance(x, Dual) else Dual(x)


def d_abs(x):
    return x if x.v >= 0 else -x  # torch.abs: gradient sign(x) (0 at 0 -- not reachable with generic inputs)


def d_sqrt(x):
    r = math.sqrt(x.v)
    return Dual(r, x.d / (2.0 * r))


def d_cos(x):
    return Dual(math.cos(x.v), -math.sin(x.v) * x.d)


def d_sin(x):
    return Dual(math.sin(x.v), math.cos(x.v) * x.d)


def d_atan2(y, x):
    den = x.v * x.v + y.v * y.v
    return Dual(math.atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / den)


def d_fmod(x, m):
    return Dual(math.fmod(x.v, m), x.d)


def corners_of(x, y, w, l, r):
    """rbbox_to_corners (odious.py:455-487): 4 (x, y) pairs, each coordinate a Dual."""
    dxcos, dxsin = w * d_cos(r) / 2.0, w * d_sin(r) / 2.0
    dycos, dysin = l * d_cos(r) / 2.0, l * d_sin(r) / 2.0
    return [(-dxcos - dysin + x, dxsin - dycos + y), (-dxcos + dysin + x, dxsin + dycos + y),
            (dxcos + dysin + x, -dxsin + dycos + y), (dxcos - dysin + x, -dxsin - dycos + y)]


def _inside(p, rect):
    """corner p inside rectangle rect (compute_vertex :33-47): projections on the two edges from corner 0"""
    ab = (rect[1][0].v - rect[0][0].v, rect[1][1].v - rect[0][1].v)
    ad = (rect[3][0].v - rect[0][0].v, rect[3][1].v - rect[0][1].v)
    ap = (p[0].v - rect[0][0].v, p[1].v - rect[0][1].v)
    abab, abap = ab[0] * ab[0] + ab[1] * ab[1], ab[0] * ap[0] + ab[1] * ap[1]
    adad, adap = ad[0] * ad[0] + ad[1] * ad[1], ad[0] * ap[0] + ad[1] * ap[1]
    return abab >= abap >= 0 and adad >= adap >= 0


def intersection_area(cg, cq):
    """rinter_area_compute (:490-503)"""
    pts = [p for p in cg if _inside(p, cq)]
    pts += [p for p in cq if _inside(p, cg)]
    for i in range(4):
        A, B = cg[i], cg[(i + 1) % 4]
        for j in range(4):
            C, D = cq[j], cq[(j + 1) % 4]
            BA = (B[0].v - A[0].v, B[1].v - A[1].v)
            CA = (C[0].v - A[0].v, C[1].v - A[1].v)
            DA = (D[0].v - A[0].v, D[1].v - A[1].v)
            acd = DA[1] * CA[0] > CA[1] * DA[0]
            bcd = (D[1].v - B[1].v) * (C[0].v - B[0].v) > (C[1].v - B[1].v) * (D[0].v - B[0].v)
            if acd == bcd:
                continue
            abc = CA[1] * BA[0] > BA[1] * CA[0]
            abd = DA[1] * BA[0] > BA[1] * DA[0]
            if abc == abd:
                continue
            if len(pts) > 7:  # :106-121: the ninth point is dropped
                continue
            ba0, ba1 = B[0] - A[0], B[1] - A[1]
            dc0, dc1 = D[0] - C[0], D[1] - C[1]
            abba = A[0] * B[1] - B[0] * A[1]
            cddc = C[0] * D[1] - D[0] * C[1]
            dh = ba1 * dc0 - ba0 * dc1
            pts.append(((abba * dc0 - ba0 * cddc) / dh, (abba * dc1 - ba1 * cddc) / dh))
    n = len(pts)
    if n < 3:
        return Dual(0.0)
    cx = sum(p[0].v for p in pts) / n
    cy = sum(p[1].v for p in pts) / n
    ang = []
    for p in pts:  # sort_vertex :300-312
        vx, vy = p[0].v - cx, p[1].v - cy
        d = math.sqrt(vx * vx + vy * vy)
        a = math.atan2(vy / d, vx / d) if d > 0 else 0.0
        ang.append(a + 2 * PI_REF if a < 0 else a)
    order = np.argsort(-np.array(ang), kind="stable")
    sp = [pts[k] for k in order]
    area = Dual(0.0)
    p1 = sp[0]
    for i in range(n - 2):  # area_polygon :352-361
        p2, p3 = sp[i + 1], sp[i + 2]
        area = area + d_abs(((p1[0] - p3[0]) * (p2[1] - p3[1]) - (p1[1] - p3[1]) * (p2[0] - p3[0])) / 2.0)
    return area


def hull_ccw(points_xy, closed=False):
    """scipy ConvexHull vertex order, as find_convex_hull (:506-518) takes it. Returns hull indices."""
    from scipy.spatial import ConvexHull
    return list(ConvexHull(np.asarray(points_xy, np.float32)).vertices)


def hull_monotone(points_xy):
    """Andrew's monotone chain (counter-clockwise from the lexicographically smallest point, collinear points dropped):
    the hull order of the DEVICE kernel, which then tries every edge (closed=True). Same area candidates as the scipy
    order; only the choice between equal-area candidates -- and with it a small part of the gradient -- can differ."""
    p = [(float(x), float(y)) for x, y in points_xy]
    idx = sorted(range(len(p)), key=lambda i: p[i])

    def cross(a, b, c):
        return (p[b][0] - p[a][0]) * (p[c][1] - p[a][1]) - (p[b][1] - p[a][1]) * (p[c][0] - p[a][0])

    h = []
    for i in idx:
        while len(h) >= 2 and cross(h[-2], h[-1], i) <= 0:
            h.pop()
        h.append(i)
    lower = len(h) + 1
    for i in reversed(idx[:-1]):
        while len(h) >= lower and cross(h[-2], h[-1], i) <= 0:
            h.pop()
        h.append(i)
    return h[:-1]


def mbr_diag(pts, closed=False, hull_fn=hull_ccw):
    """mbr_diag_convex_hull (:596-626) on the hull of the 8 corner points (Duals). closed=True also tries the closing edge
    (what an implementation that does not depend on Qhull's start vertex does)."""
    idx = hull_fn([(p[0].v, p[1].v) for p in pts])
    hp = [pts[k] for k in idx]
    n = len(hp)
    best = None
    for e in range(n if closed else n - 1):
        a, b = hp[e], hp[(e + 1) % n]
        th = d_abs(d_fmod(d_atan2(b[1] - a[1], b[0] - a[0]), PI_REF / 2.0))
        c, s, ns = d_cos(th), d_cos(th - PI_REF / 2.0), d_cos(th + PI_REF / 2.0)
        rx = [c * p[0] + s * p[1] for p in hp]
        ry = [ns * p[0] + c * p[1] for p in hp]
        xs, ys = [r.v for r in rx], [r.v for r in ry]
        ex = rx[int(np.argmax(xs))] - rx[int(np.argmin(xs))]
        ey = ry[int(np.argmax(ys))] - ry[int(np.argmin(ys))]
        area = ex.v * ey.v
        if best is None or area < best[0]:
            best = (area, ex, ey)
    return d_sqrt(best[1] * best[1] + best[2] * best[2])


def odiou_term(g, q, closed=False, device_convention=False):
    """One pair: returns (term, d term / d q[7]); the loss is 2 * sum(w_i * term_i) / batch_size (:895-899).
    device_convention=True: monotone-chain hull + all edges, the kernel's choice (see hull_monotone)."""
    g = [float(v) for v in g]
    if not (g[3] > 0 and g[4] > 0 and g[5] > 0 and q[3] > 0 and q[4] > 0 and q[5] > 0):
        return 0.0, np.zeros(ND)
    gc = [min(max(v, -200.0), 200.0) for v in g]
    qd = []
    for i in range(ND):  # torch.clamp: identity gradient inside [-200, 200], zero outside
        v = float(q[i])
        qd.append(Dual.var(v, i) if -200.0 <= v <= 200.0 else Dual(min(max(v, -200.0), 200.0)))
    angle = 1.25 * (1.0 - d_abs(d_cos(qd[6] - gc[6])))
    cg = corners_of(Dual(gc[0]), Dual(gc[1]), Dual(gc[3]), Dual(gc[4]), Dual(gc[6]))
    cq = corners_of(qd[0], qd[1], qd[3], qd[4], qd[6])
    inter_area = intersection_area(cg, cq)
    dist2 = sum(((gc[i] - qd[i]) * (gc[i] - qd[i]) for i in range(3)), Dual(0.0))
    diag_bev = mbr_diag(cg + cq, closed=True, hull_fn=hull_monotone) if device_convention else mbr_diag(cg + cq, closed=closed)
    top_g, bot_g = gc[2] + 0.5 * gc[5], gc[2] - 0.5 * gc[5]
    top_q, bot_q = qd[2] + 0.5 * qd[5], qd[2] - 0.5 * qd[5]
    # torch.min / torch.max of two tensors: on an exact tie the gradient is split evenly between the operands
    top = top_q if top_q.v < top_g else (Dual(top_g) if top_q.v > top_g else Dual(top_g, 0.5 * top_q.d))
    bot = bot_q if bot_q.v > bot_g else (Dual(bot_g) if bot_q.v < bot_g else Dual(bot_g, 0.5 * bot_q.d))
    inter_h = top - bot
    if inter_h.v < 0:
        inter_h = Dual(0.0)                                 # :880 in-place zeroing cuts the gradient
    diag3d2 = diag_bev * diag_bev + inter_h * inter_h + 1e-7
    vol_g = gc[3] * gc[4] * gc[5]
    vol_q = qd[3] * qd[4] * qd[5]
    inc = inter_h * inter_area
    iou = inc / (vol_g + vol_q - inc)
    term = 1.0 - iou + dist2 / diag3d2 + angle
    return term.v, term.d


def odiou_loss(gboxes, qboxes, weights, batch_size, closed=False, device_convention=False):
    """odiou_3D.forward: returns (loss, d loss / d qboxes (N,7), per-pair terms (N,))."""
    n = len(gboxes)
    terms, grads = np.zeros(n), np.zeros((n, ND))
    for i in range(n):
        terms[i], grads[i] = odiou_term(gboxes[i], qboxes[i], closed=closed, device_convention=device_convention)
    w = np.asarray(weights, np.float64)
    return 2.0 * float((terms * w).sum()) / batch_size, 2.0 * grads * w[:, None] / batch_size, terms
