// Device core of the ODIoU term (csrc/odiou.hip describes the algorithm and cites the reference): float64 forward-mode duals,
// one (target, prediction) pair per thread. Shared by odiou.hip (the standalone op) and head_loss.hip (the fused SE-SSD loss).
#pragma once
#include "common.hpp"

namespace {

constexpr int NB = 7;   // box parameters [x, y, z, w, l, h, r]
constexpr int ND = 1;   // partial derivatives carried per LANE: lane c of a pair's group of eight differentiates with respect to box
                        // parameter c (round 4: one thread carrying all seven was a 140 us serial float64 chain; per component the
                        // formulas -- and therefore the bits -- are the same)
constexpr double PI_REF = 3.1415926;

struct Dual {
  double v;
  double d[ND];
};

__device__ __forceinline__ Dual dconst(double v) {
  Dual r;
  r.v = v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = 0.0;
  return r;
}
__device__ __forceinline__ Dual dvar(double v) {   // the variable this lane differentiates with respect to
  Dual r = dconst(v);
  r.d[0] = 1.0;
  return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator-(const Dual& a) {
  Dual r;
  r.v = -a.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i];
  return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + b.d[i] * a.v;
  return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, double s) {
  Dual r;
  r.v = a.v * s;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s;
  return r;
}
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v / b.v;
  const double inv = 1.0 / (b.v * b.v);
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] * b.v - b.d[i] * a.v) * inv;
  return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, double s) {
  Dual r = a;
  r.v += s;
  return r;
}
__device__ __forceinline__ Dual dabs(const Dual& a) { return a.v >= 0.0 ? a : -a; }
__device__ __forceinline__ Dual dsqrt(const Dual& a) {
  Dual r;
  r.v = sqrt(a.v);
  const double s = 0.5 / r.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s;
  return r;
}
__device__ __forceinline__ Dual dcos(const Dual& a) {
  Dual r;
  r.v = cos(a.v);
  const double s = -sin(a.v);
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s;
  return r;
}
__device__ __forceinline__ Dual dsin(const Dual& a) {
  Dual r;
  r.v = sin(a.v);
  const double s = cos(a.v);
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s;
  return r;
}
__device__ __forceinline__ Dual datan2(const Dual& y, const Dual& x) {
  Dual r;
  r.v = atan2(y.v, x.v);
  const double inv = 1.0 / (x.v * x.v + y.v * y.v);
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * inv;
  return r;
}

struct P2 {
  Dual x, y;
};

// rbbox_to_corners (odious.py:455-487)
__device__ void corners_of(const Dual& cx, const Dual& cy, const Dual& w, const Dual& l, const Dual& r, P2* c) {
  const Dual co = dcos(r), si = dsin(r);
  const Dual dxcos = w * co * 0.5, dxsin = w * si * 0.5, dycos = l * co * 0.5, dysin = l * si * 0.5;
  c[0].x = -dxcos - dysin + cx; c[0].y = dxsin - dycos + cy;
  c[1].x = -dxcos + dysin + cx; c[1].y = dxsin + dycos + cy;
  c[2].x = dxcos + dysin + cx;  c[2].y = -dxsin + dycos + cy;
  c[3].x = dxcos - dysin + cx;  c[3].y = -dxsin - dycos + cy;
}

__device__ bool inside_rect(const P2& p, const P2* rc) {  // compute_vertex :33-47
  const double ab0 = rc[1].x.v - rc[0].x.v, ab1 = rc[1].y.v - rc[0].y.v;
  const double ad0 = rc[3].x.v - rc[0].x.v, ad1 = rc[3].y.v - rc[0].y.v;
  const double ap0 = p.x.v - rc[0].x.v, ap1 = p.y.v - rc[0].y.v;
  const double abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  const double adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0.0 && adad >= adap && adap >= 0.0;
}

__device__ Dual intersection_area(const P2* cg, const P2* cq) {
  P2 pts[8];
  int n = 0;
  for (int i = 0; i < 4; ++i)
    if (inside_rect(cg[i], cq)) pts[n++] = cg[i];
  for (int i = 0; i < 4; ++i)
    if (inside_rect(cq[i], cg) && n < 8) pts[n++] = cq[i];
  for (int i = 0; i < 4; ++i) {
    const P2 &A = cg[i], &B = cg[(i + 1) & 3];
    for (int j = 0; j < 4; ++j) {
      const P2 &C = cq[j], &D = cq[(j + 1) & 3];
      const double BA0 = B.x.v - A.x.v, BA1 = B.y.v - A.y.v, CA0 = C.x.v - A.x.v, CA1 = C.y.v - A.y.v;
      const double DA0 = D.x.v - A.x.v, DA1 = D.y.v - A.y.v;
      const bool acd = DA1 * CA0 > CA1 * DA0;
      const bool bcd = (D.y.v - B.y.v) * (C.x.v - B.x.v) > (C.y.v - B.y.v) * (D.x.v - B.x.v);
      if (acd == bcd) continue;
      const bool abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
      if (abc == abd) continue;
      if (n > 7) continue;  // :106-121 the ninth point is dropped
      const Dual ba0 = B.x - A.x, ba1 = B.y - A.y, dc0 = D.x - C.x, dc1 = D.y - C.y;
      const Dual abba = A.x * B.y - B.x * A.y, cddc = C.x * D.y - D.x * C.y;
      const Dual dh = ba1 * dc0 - ba0 * dc1;
      pts[n].x = (abba * dc0 - ba0 * cddc) / dh;
      pts[n].y = (abba * dc1 - ba1 * cddc) / dh;
      ++n;
    }
  }
  if (n < 3) return dconst(0.0);
  double cx = 0.0, cy = 0.0;
  for (int i = 0; i < n; ++i) { cx += pts[i].x.v; cy += pts[i].y.v; }
  cx /= n; cy /= n;
  double ang[8];
  int ord[8];
  for (int i = 0; i < n; ++i) {  // sort_vertex :300-312: descending angle around the centroid
    const double vx = pts[i].x.v - cx, vy = pts[i].y.v - cy;
    const double d = sqrt(vx * vx + vy * vy);
    double a = d > 0.0 ? atan2(vy / d, vx / d) : 0.0;
    if (a < 0.0) a += 2.0 * PI_REF;
    ang[i] = a;
    ord[i] = i;
  }
  for (int i = 1; i < n; ++i) {  // stable insertion sort, descending
    const int oi = ord[i];
    int k = i - 1;
    while (k >= 0 && ang[ord[k]] < ang[oi]) { ord[k + 1] = ord[k]; --k; }
    ord[k + 1] = oi;
  }
  Dual area = dconst(0.0);
  const P2& p1 = pts[ord[0]];
  for (int i = 0; i < n - 2; ++i) {  // area_polygon :352-361
    const P2 &p2 = pts[ord[i + 1]], &p3 = pts[ord[i + 2]];
    area = area + dabs(((p1.x - p3.x) * (p2.y - p3.y) - (p1.y - p3.y) * (p2.x - p3.x)) * 0.5);
  }
  return area;
}

// Andrew's monotone chain on the values; returns the hull (counter-clockwise, starting at the lexicographically smallest
// point) as indices into pts. Collinear points are dropped (cross <= 0 pops).
__device__ int hull_monotone(const P2* pts, int* hull) {
  int idx[8];
  for (int i = 0; i < 8; ++i) idx[i] = i;
  for (int i = 1; i < 8; ++i) {  // sort by (x, y)
    const int t = idx[i];
    int k = i - 1;
    while (k >= 0 && (pts[idx[k]].x.v > pts[t].x.v || (pts[idx[k]].x.v == pts[t].x.v && pts[idx[k]].y.v > pts[t].y.v))) {
      idx[k + 1] = idx[k];
      --k;
    }
    idx[k + 1] = t;
  }
  int h[16];
  int m = 0;
  for (int i = 0; i < 8; ++i) {
    while (m >= 2) {
      const P2 &a = pts[h[m - 2]], &b = pts[h[m - 1]], &c = pts[idx[i]];
      if ((b.x.v - a.x.v) * (c.y.v - a.y.v) - (b.y.v - a.y.v) * (c.x.v - a.x.v) <= 0.0) --m; else break;
    }
    h[m++] = idx[i];
  }
  const int lower = m + 1;
  for (int i = 6; i >= 0; --i) {
    while (m >= lower) {
      const P2 &a = pts[h[m - 2]], &b = pts[h[m - 1]], &c = pts[idx[i]];
      if ((b.x.v - a.x.v) * (c.y.v - a.y.v) - (b.y.v - a.y.v) * (c.x.v - a.x.v) <= 0.0) --m; else break;
    }
    h[m++] = idx[i];
  }
  --m;  // the last point repeats the first
  for (int i = 0; i < m; ++i) hull[i] = h[i];
  return m;
}

// mbr_diag_convex_hull (:596-626) over every hull edge
__device__ Dual mbr_diag(const P2* pts) {
  int hull[8];
  const int n = hull_monotone(pts, hull);
  double best_area = 1e300;
  Dual bex = dconst(0.0), bey = dconst(0.0);
  for (int e = 0; e < n; ++e) {
    const P2 &a = pts[hull[e]], &b = pts[hull[(e + 1) % n]];
    Dual th = datan2(b.y - a.y, b.x - a.x);
    th.v = fmod(th.v, PI_REF / 2.0);  // derivative of fmod with respect to its first argument is 1
    th = dabs(th);
    const Dual c = dcos(th), s = dcos(th + (-PI_REF / 2.0)), ns = dcos(th + (PI_REF / 2.0));
    int ixmin = 0, ixmax = 0, iymin = 0, iymax = 0;
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    for (int k = 0; k < n; ++k) {
      const P2& p = pts[hull[k]];
      const double rx = c.v * p.x.v + s.v * p.y.v, ry = ns.v * p.x.v + c.v * p.y.v;
      if (k == 0 || rx < xmin) { xmin = rx; ixmin = k; }
      if (k == 0 || rx > xmax) { xmax = rx; ixmax = k; }
      if (k == 0 || ry < ymin) { ymin = ry; iymin = k; }
      if (k == 0 || ry > ymax) { ymax = ry; iymax = k; }
    }
    const double area = (xmax - xmin) * (ymax - ymin);
    if (area < best_area) {
      best_area = area;
      const P2 &pa = pts[hull[ixmax]], &pb = pts[hull[ixmin]], &pc = pts[hull[iymax]], &pd = pts[hull[iymin]];
      bex = (c * pa.x + s * pa.y) - (c * pb.x + s * pb.y);
      bey = (ns * pc.x + c * pc.y) - (ns * pd.x + c * pd.y);
    }
  }
  return dsqrt(bex * bex + bey * bey);
}

// term and d term / d q[comp] of ONE pair: g, qv = [x, y, z, w, l, h, r] (float64 copies of the float32 boxes), comp in 0..6 the
// box parameter this lane differentiates with respect to. odious.py:851-899.
__device__ void odiou_eval(const double* g_in, const double* qv, int comp, double* term_out, double* grad_out) {
  double g[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) g[k] = g_in[k];
  if (!(g[3] > 0 && g[4] > 0 && g[5] > 0 && qv[3] > 0 && qv[4] > 0 && qv[5] > 0)) {  // :851-853 indicator
    *term_out = 0.0;
    *grad_out = 0.0;
    return;
  }
  Dual q[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {  // torch.clamp(-200, 200) :855-856: identity gradient inside, zero outside
    g[k] = fmin(fmax(g[k], -200.0), 200.0);
    q[k] = dconst(fmin(fmax(qv[k], -200.0), 200.0));
    if (k == comp && qv[k] >= -200.0 && qv[k] <= 200.0) q[k].d[0] = 1.0;
  }
  const Dual angle = (dconst(1.0) - dabs(dcos(q[6] + (-g[6])))) * 1.25;
  P2 c[8];
  corners_of(dconst(g[0]), dconst(g[1]), dconst(g[3]), dconst(g[4]), dconst(g[6]), c);
  corners_of(q[0], q[1], q[3], q[4], q[6], c + 4);
  const Dual inter_area = intersection_area(c, c + 4);
  Dual dist2 = dconst(0.0);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const Dual df = dconst(g[k]) - q[k];
    dist2 = dist2 + df * df;
  }
  const Dual diag = mbr_diag(c);
  const double top_g = g[2] + 0.5 * g[5], bot_g = g[2] - 0.5 * g[5];
  const Dual top_q = q[2] + q[5] * 0.5, bot_q = q[2] - q[5] * 0.5;
  // torch.min / torch.max of two tensors: on an exact tie the gradient is split evenly between the operands
  Dual top = top_q.v < top_g ? top_q : (top_q.v > top_g ? dconst(top_g) : top_q * 0.5 + 0.5 * top_g);
  Dual bot = bot_q.v > bot_g ? bot_q : (bot_q.v < bot_g ? dconst(bot_g) : bot_q * 0.5 + 0.5 * bot_g);
  Dual inter_h = top - bot;
  if (inter_h.v < 0.0) inter_h = dconst(0.0);  // :880
  const Dual diag3d2 = diag * diag + inter_h * inter_h + 1e-7;
  const double vol_g = g[3] * g[4] * g[5];
  const Dual vol_q = q[3] * q[4] * q[5];
  const Dual inc = inter_h * inter_area;
  const Dual iou = inc / (vol_q + vol_g - inc);
  const Dual term = dconst(1.0) - iou + dist2 / diag3d2 + angle;
  *term_out = term.v;
  *grad_out = term.d[0];
}

}  // namespace
