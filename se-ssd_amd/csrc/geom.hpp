// Device geometry shared by iou3d.hip, nms.hip and postprocess.hip (gfx950).
// Two rotated-rectangle intersection routines live here because the reference has two:
//   * rect_overlap_f32  - the iou3d_cuda algorithm (det3d/core/iou3d/src/iou3d_kernel.cu:125-245):
//       corners rotated about the centre, 16 edge/edge crossing tests, 8 corner-in-rect tests
//       (margin 1e-5), stable angular sort about the centroid, shoelace fan. float32 throughout,
//       same operation order as the reference so results agree to trig-function rounding.
//   * quad_clip_area    - polygon clipping for the boost::geometry based CPU rotated NMS that
//       MultiGroupHead.predict really calls (det3d/ops/nms/nms_cpu.h:72-168). boost is a black
//       box; this is Sutherland-Hodgman in float64 (MI355X runs f64 vector math at the f32 rate).
#pragma once
#include "common.hpp"

#define SESSD_IOU_EPS 1e-8f
#define SESSD_IOU_MARGIN 1e-5f
#define SESSD_IOU_MAXPTS 16  // the reference's cross_points[16]

struct sessd_pt {
  float x, y;
};

static __device__ __forceinline__ float sessd_cross3(sessd_pt p1, sessd_pt p2, sessd_pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

static __device__ __forceinline__ sessd_pt sessd_rot_about(sessd_pt c, float co, float si, sessd_pt p) {
  sessd_pt r;
  r.x = (p.x - c.x) * co + (p.y - c.y) * si + c.x;
  r.y = -(p.x - c.x) * si + (p.y - c.y) * co + c.y;
  return r;
}

// Rotated corners of one [x1,y1,x2,y2,angle] rectangle, computed once per box.
struct sessd_rect {
  float x1, y1, x2, y2, ang;
  sessd_pt c[4];
  float cx, cy, rad;  // centre and half diagonal (for the exact-zero early out)
};

static __device__ __forceinline__ void sessd_rect_init(sessd_rect& R, float x1, float y1, float x2, float y2,
                                                       float ang) {
  R.x1 = x1; R.y1 = y1; R.x2 = x2; R.y2 = y2; R.ang = ang;
  sessd_pt ctr = {(x1 + x2) / 2, (y1 + y2) / 2};
  float co = cosf(ang), si = sinf(ang);
  sessd_pt p0 = {x1, y1}, p1 = {x2, y1}, p2 = {x2, y2}, p3 = {x1, y2};
  R.c[0] = sessd_rot_about(ctr, co, si, p0);
  R.c[1] = sessd_rot_about(ctr, co, si, p1);
  R.c[2] = sessd_rot_about(ctr, co, si, p2);
  R.c[3] = sessd_rot_about(ctr, co, si, p3);
  R.cx = ctr.x; R.cy = ctr.y;
  float hx = (x2 - x1) * 0.5f, hy = (y2 - y1) * 0.5f;
  R.rad = sqrtf(hx * hx + hy * hy);
}

static __device__ __forceinline__ bool sessd_in_rect(const sessd_rect& R, sessd_pt p) {
  // rotate the point back by -angle about the centre, then an axis-aligned test with margin
  float cx = (R.x1 + R.x2) / 2, cy = (R.y1 + R.y2) / 2;
  float c = cosf(-R.ang), s = sinf(-R.ang);
  float rx = (p.x - cx) * c + (p.y - cy) * s + cx;
  float ry = -(p.x - cx) * s + (p.y - cy) * c + cy;
  return rx > R.x1 - SESSD_IOU_MARGIN && rx < R.x2 + SESSD_IOU_MARGIN && ry > R.y1 - SESSD_IOU_MARGIN &&
         ry < R.y2 + SESSD_IOU_MARGIN;
}

static __device__ __forceinline__ bool sessd_seg_cross(sessd_pt p1, sessd_pt p0, sessd_pt q1, sessd_pt q0,
                                                       sessd_pt* ans) {
  bool touch = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
               fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!touch) return false;
  float s1 = sessd_cross3(q0, p1, p0), s2 = sessd_cross3(p1, q1, p0);
  float s3 = sessd_cross3(p0, q1, q0), s4 = sessd_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  float s5 = sessd_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > SESSD_IOU_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

// Per-thread point list kept in LDS, k-major ([k][thread]) so that the 64 lanes of a wave hit 64
// consecutive dwords whatever k is: dynamic indexing without scratch memory and without conflicts.
struct sessd_ptlist {
  float* x;
  float* y;
  float* a;
  int stride;  // threads per block
};

// Overlap area of two rotated rectangles, iou3d_cuda semantics.
static __device__ float sessd_rect_overlap_f32(const sessd_rect& A, const sessd_rect& B, sessd_ptlist L) {
  // Exactly zero in the reference too: no edge can cross and no corner can lie inside (+margin).
  {
    float dx = A.cx - B.cx, dy = A.cy - B.cy;
    float reach = A.rad + B.rad + 1e-2f;
    if (dx * dx + dy * dy > reach * reach) return 0.f;
  }
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sessd_pt ans;
      if (sessd_seg_cross(A.c[(i + 1) & 3], A.c[i], B.c[(j + 1) & 3], B.c[j], &ans) && cnt < SESSD_IOU_MAXPTS) {
        sx += ans.x; sy += ans.y;
        L.x[cnt * L.stride] = ans.x; L.y[cnt * L.stride] = ans.y;
        ++cnt;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (sessd_in_rect(A, B.c[k]) && cnt < SESSD_IOU_MAXPTS) {
      sx += B.c[k].x; sy += B.c[k].y;
      L.x[cnt * L.stride] = B.c[k].x; L.y[cnt * L.stride] = B.c[k].y;
      ++cnt;
    }
    if (sessd_in_rect(B, A.c[k]) && cnt < SESSD_IOU_MAXPTS) {
      sx += A.c[k].x; sy += A.c[k].y;
      L.x[cnt * L.stride] = A.c[k].x; L.y[cnt * L.stride] = A.c[k].y;
      ++cnt;
    }
  }
  if (cnt < 3) return 0.f;  // fan of fewer than 3 points has zero area (reference loops do not run / sum 0)
  float ctx = sx / cnt, cty = sy / cnt;
  for (int k = 0; k < cnt; ++k) L.a[k * L.stride] = atan2f(L.y[k * L.stride] - cty, L.x[k * L.stride] - ctx);
  // stable insertion sort by angle == the reference's bubble sort with strict '>' swaps
  for (int k = 1; k < cnt; ++k) {
    float ak = L.a[k * L.stride], xk = L.x[k * L.stride], yk = L.y[k * L.stride];
    int m = k - 1;
    while (m >= 0 && L.a[m * L.stride] > ak) {
      L.a[(m + 1) * L.stride] = L.a[m * L.stride];
      L.x[(m + 1) * L.stride] = L.x[m * L.stride];
      L.y[(m + 1) * L.stride] = L.y[m * L.stride];
      --m;
    }
    L.a[(m + 1) * L.stride] = ak; L.x[(m + 1) * L.stride] = xk; L.y[(m + 1) * L.stride] = yk;
  }
  float area = 0.f;
  float x0 = L.x[0], y0 = L.y[0];
  for (int k = 0; k < cnt - 1; ++k) {
    float ux = L.x[k * L.stride] - x0, uy = L.y[k * L.stride] - y0;
    float vx = L.x[(k + 1) * L.stride] - x0, vy = L.y[(k + 1) * L.stride] - y0;
    area += ux * vy - uy * vx;
  }
  return fabsf(area) / 2.0f;
}

// ---------------------------------------------------------------------------------------------
// rotate_nms (predict path) geometry: corners as numpy builds them
//   det3d/core/bbox/box_np_ops.py:267-294 (corners_nd, origin 0.5), :433-446 (rotation_2d), :512-532
// det = [x, y, w, l, r]; corner k = R(r) * (u_k * (w,l)) + (x,y), u = (-.5,-.5),(-.5,.5),(.5,.5),(.5,-.5)
static __device__ __forceinline__ void sessd_box2d_corners(const float* det, float* c8) {
  const float ux[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, uy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
  float s = sinf(det[4]), c = cosf(det[4]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float px = det[2] * ux[k], py = det[3] * uy[k];
    float a = px * c, b = py * s, d = px * (-s), e = py * c;
    c8[2 * k] = (a + b) + det[0];
    c8[2 * k + 1] = (d + e) + det[1];
  }
}

static __device__ __forceinline__ double sessd_poly_area2(const double* px, const double* py, int n) {
  double a = 0;
  for (int i = 0; i < n; ++i) {
    int j = i + 1 == n ? 0 : i + 1;
    a += px[i] * py[j] - px[j] * py[i];
  }
  return a;  // twice the signed area
}

// Intersection area of two convex quads (either orientation); P, Q are (4,2) float32 corner sets.
static __device__ double sessd_quad_clip_area(const float* P, const float* Q) {
  double sx[12], sy[12], tx[12], ty[12], qx[4], qy[4];
  int ns = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sx[i] = P[2 * i]; sy[i] = P[2 * i + 1];
    qx[i] = Q[2 * i]; qy[i] = Q[2 * i + 1];
  }
  double ca = sessd_poly_area2(qx, qy, 4);
  if (ca == 0) return 0;
  const double sgn = ca > 0 ? 1.0 : -1.0;
  for (int e = 0; e < 4 && ns > 0; ++e) {
    double ax = qx[e], ay = qy[e], bx = qx[(e + 1) & 3], by = qy[(e + 1) & 3];
    int nt = 0;
    for (int i = 0; i < ns; ++i) {
      int j = i + 1 == ns ? 0 : i + 1;
      double cx = sx[i], cy = sy[i], dx = sx[j], dy = sy[j];
      double sc = sgn * ((bx - ax) * (cy - ay) - (by - ay) * (cx - ax));
      double sd = sgn * ((bx - ax) * (dy - ay) - (by - ay) * (dx - ax));
      if (sc >= 0) { tx[nt] = cx; ty[nt] = cy; ++nt; }
      if ((sc >= 0) != (sd >= 0)) {
        double t = sc / (sc - sd);
        tx[nt] = cx + t * (dx - cx); ty[nt] = cy + t * (dy - cy);
        ++nt;
      }
    }
    ns = nt;
    for (int i = 0; i < ns; ++i) { sx[i] = tx[i]; sy[i] = ty[i]; }
  }
  if (ns < 3) return 0;
  return fabs(sessd_poly_area2(sx, sy, ns)) * 0.5;
}

// Intersection area of two convex quads WITHOUT any indexed scratch: by Green's theorem the boundary of P n Q is
// made of the parts of P's edges inside Q and the parts of Q's edges inside P, so
//     area = 1/2 * sum over those oriented pieces of cross(start, end).
// Each edge is clipped against the other quad parametrically (Cyrus-Beck, 4 half-planes), everything stays in
// registers (float64) and is fully unrolled. Both quads are first made counter-clockwise. An edge of Q lying ON an
// edge of P is counted once (closed test for P's edges, open test for Q's), so duplicate boxes give |P|.
static __device__ __forceinline__ double sessd_clip_piece(double ax, double ay, double bx, double by, const double* qx,
                                                          const double* qy, bool strict) {
  const double dx = bx - ax, dy = by - ay;
  double t0 = 0.0, t1 = 1.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double ex = qx[(k + 1) & 3] - qx[k], ey = qy[(k + 1) & 3] - qy[k];
    const double c = ex * (ay - qy[k]) - ey * (ax - qx[k]);  // >= 0 : a is inside half-plane k
    const double sl = ex * dy - ey * dx;                      // d/dt of that quantity along a->b
    if (sl > 0.0) {
      t0 = fmax(t0, -c / sl);
    } else if (sl < 0.0) {
      t1 = fmin(t1, -c / sl);
    } else {
      ok = ok && (strict ? c > 0.0 : c >= 0.0);
    }
  }
  if (!ok || !(t0 < t1)) return 0.0;
  const double sx = ax + t0 * dx, sy = ay + t0 * dy, ex2 = ax + t1 * dx, ey2 = ay + t1 * dy;
  return sx * ey2 - sy * ex2;
}

static __device__ double sessd_quad_inter_area_green(const float* P, const float* Q) {
  double px[4], py[4], qx[4], qy[4];
  const double ox = P[0], oy = P[1];  // translate: keeps the cross products small
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    px[i] = (double)P[2 * i] - ox; py[i] = (double)P[2 * i + 1] - oy;
    qx[i] = (double)Q[2 * i] - ox; qy[i] = (double)Q[2 * i + 1] - oy;
  }
  if (sessd_poly_area2(px, py, 4) < 0) { double t = px[1]; px[1] = px[3]; px[3] = t; t = py[1]; py[1] = py[3]; py[3] = t; }
  if (sessd_poly_area2(qx, qy, 4) < 0) { double t = qx[1]; qx[1] = qx[3]; qx[3] = t; t = qy[1]; qy[1] = qy[3]; qy[3] = t; }
  double a2 = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a2 += sessd_clip_piece(px[i], py[i], px[(i + 1) & 3], py[(i + 1) & 3], qx, qy, false);
    a2 += sessd_clip_piece(qx[i], qy[i], qx[(i + 1) & 3], qy[(i + 1) & 3], px, py, true);
  }
  return a2 > 0 ? 0.5 * a2 : 0.0;
}
