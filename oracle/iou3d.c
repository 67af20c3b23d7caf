/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Plain-C restatement of the reference's rotated BEV / 3-D overlap
 *   det3d/core/iou3d/src/iou3d_cpu.cpp:36-245  (cross, check_rect_cross, check_in_box2d, intersection, box_overlap)
 *   det3d/core/iou3d/src/iou3d_cpu.cpp:248-336 (iou_bev, the three *_cpu entry points)
 *   det3d/core/iou3d/src/iou3d_kernel.cu:247-268 (iou_3d device variant: returns 0 when z ranges do not overlap)
 *   det3d/core/iou3d/src/iou3d_kernel.cu:414-422 (iou_normal)
 * Algorithm: rotate the four corners of both rectangles, collect the (up to 16)
 * edge/edge crossing points and the (up to 8) corners lying inside the other
 * rectangle (1e-5 margin), order them by atan2 around their centroid with a bubble
 * sort, shoelace the fan.
 *
 * Pinned against the compiled reference source (oracle/_ref, built from
 * /root/reference/det3d/core/iou3d/src/iou3d_cpu.cpp) and against the golden
 * vectors that binary produced (tests/golden/iou3d_ref.npz).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define O_EPS 1e-8f
#define O_MARGIN 1e-5f

typedef struct { float x, y; } pt;

static float cr2(pt a, pt b) { return a.x * b.y - a.y * b.x; }
static float cr3(pt p1, pt p2, pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static float mn(float a, float b) { return a < b ? a : b; }
static float mx(float a, float b) { return a > b ? a : b; }

static int bbox_touch(pt p1, pt p2, pt q1, pt q2) {
  return mn(p1.x, p2.x) <= mx(q1.x, q2.x) && mn(q1.x, q2.x) <= mx(p1.x, p2.x) && mn(p1.y, p2.y) <= mx(q1.y, q2.y) &&
         mn(q1.y, q2.y) <= mx(p1.y, p2.y);
}

/* x1,y1,x2,y2,angle */
static int inside_rect(float x1, float y1, float x2, float y2, float ang, pt p) {
  float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2;
  float c = cosf(-ang), s = sinf(-ang);
  float rx = (p.x - cx) * c + (p.y - cy) * s + cx;
  float ry = -(p.x - cx) * s + (p.y - cy) * c + cy;
  return rx > x1 - O_MARGIN && rx < x2 + O_MARGIN && ry > y1 - O_MARGIN && ry < y2 + O_MARGIN;
}

static int seg_cross(pt p1, pt p0, pt q1, pt q0, pt *ans) {
  if (!bbox_touch(p0, p1, q0, q1)) return 0;
  float s1 = cr3(q0, p1, p0), s2 = cr3(p1, q1, p0), s3 = cr3(p0, q1, q0), s4 = cr3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cr3(q1, p1, p0);
  if (fabsf(s5 - s1) > O_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static pt rot(pt c, float co, float si, pt p) {
  pt r;
  r.x = (p.x - c.x) * co + (p.y - c.y) * si + c.x;
  r.y = -(p.x - c.x) * si + (p.y - c.y) * co + c.y;
  return r;
}

/* rectangles given as x1,y1,x2,y2,angle */
float oracle_rect_overlap(float ax1, float ay1, float ax2, float ay2, float aa, float bx1, float by1, float bx2,
                          float by2, float ba) {
  /* the reference's Point(double,double) ctor: centre computed in float, stored float */
  pt ca = {(ax1 + ax2) / 2, (ay1 + ay2) / 2}, cb = {(bx1 + bx2) / 2, (by1 + by2) / 2};
  pt A[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}}, B[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}};
  float aco = cosf(aa), asi = sinf(aa), bco = cosf(ba), bsi = sinf(ba);
  for (int k = 0; k < 4; ++k) {
    A[k] = rot(ca, aco, asi, A[k]);
    B[k] = rot(cb, bco, bsi, B[k]);
  }
  A[4] = A[0];
  B[4] = B[0];
  pt P[24], ctr = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_cross(A[i + 1], A[i], B[j + 1], B[j], &P[cnt])) {
        ctr.x += P[cnt].x;
        ctr.y += P[cnt].y;
        cnt++;
      }
  for (int k = 0; k < 4; ++k) {
    if (inside_rect(ax1, ay1, ax2, ay2, aa, B[k])) {
      ctr.x += B[k].x; ctr.y += B[k].y; P[cnt++] = B[k];
    }
    if (inside_rect(bx1, by1, bx2, by2, ba, A[k])) {
      ctr.x += A[k].x; ctr.y += A[k].y; P[cnt++] = A[k];
    }
  }
  ctr.x /= cnt; /* cnt == 0 -> NaN centre, harmless: the loops below do not run */
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(P[i].y - ctr.y, P[i].x - ctr.x) > atan2f(P[i + 1].y - ctr.y, P[i + 1].x - ctr.x)) {
        pt t = P[i]; P[i] = P[i + 1]; P[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; ++k) {
    pt u = {P[k].x - P[0].x, P[k].y - P[0].y}, v = {P[k + 1].x - P[0].x, P[k + 1].y - P[0].y};
    area += cr2(u, v);
  }
  return fabsf(area) / 2.0f;
}

static float bev_iou5(const float *a, const float *b) {
  float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  float so = oracle_rect_overlap(a[0], a[1], a[2], a[3], a[4], b[0], b[1], b[2], b[3], b[4]);
  return so / fmaxf(sa + sb - so, O_EPS);
}

/* (N,5)x(M,5) -> (N,M) overlap area ; boxes [x1,y1,x2,y2,ry]   (iou3d_cpu.cpp:258-281) */
void oracle_boxes_overlap_bev(const float *a, int N, const float *b, int M, float *out) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j)
      out[(size_t)i * M + j] = oracle_rect_overlap(a[5 * i], a[5 * i + 1], a[5 * i + 2], a[5 * i + 3], a[5 * i + 4],
                                                   b[5 * j], b[5 * j + 1], b[5 * j + 2], b[5 * j + 3], b[5 * j + 4]);
}

/* aligned pairs (iou3d_kernel.cu:284-293) */
void oracle_boxes_aligned_overlap_bev(const float *a, const float *b, int N, float *out) {
  for (int i = 0; i < N; ++i)
    out[i] = oracle_rect_overlap(a[5 * i], a[5 * i + 1], a[5 * i + 2], a[5 * i + 3], a[5 * i + 4], b[5 * i],
                                 b[5 * i + 1], b[5 * i + 2], b[5 * i + 3], b[5 * i + 4]);
}

/* (iou3d_cpu.cpp:284-304) */
void oracle_boxes_iou_bev(const float *a, int N, const float *b, int M, float *out) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) out[(size_t)i * M + j] = bev_iou5(a + 5 * i, b + 5 * j);
}

/* boxes [x1,y1,z1,x2,y2,z2,ry]. gpu_variant=1 follows iou3d_kernel.cu:256-268 (0 when the
 * z ranges do not overlap); gpu_variant=0 follows iou3d_cpu.cpp:306-336 (no early return,
 * so a clamped 1e-8 height is used; the stray write at :328-330 is overwritten and has no
 * visible effect when N==M or the index stays in range). */
void oracle_boxes_iou3d(const float *a, int N, const float *b, int M, float *out, int gpu_variant) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const float *p = a + 7 * i, *q = b + 7 * j;
      float va = (p[3] - p[0]) * (p[4] - p[1]) * (p[5] - p[2]);
      float vb = (q[3] - q[0]) * (q[4] - q[1]) * (q[5] - q[2]);
      float lo = fmaxf(p[2], q[2]), hi = fminf(p[5], q[5]);
      float dh = fmaxf(hi - lo, O_EPS);
      if (gpu_variant && dh == O_EPS) {
        out[(size_t)i * M + j] = 0.f;
        continue;
      }
      float vo = oracle_rect_overlap(p[0], p[1], p[3], p[4], p[6], q[0], q[1], q[3], q[4], q[6]) * dh;
      out[(size_t)i * M + j] = vo / fmaxf(va + vb - vo, O_EPS);
    }
}

static float normal_iou5(const float *a, const float *b) {
  float l = fmaxf(a[0], b[0]), r = fminf(a[2], b[2]), t = fmaxf(a[1], b[1]), bt = fminf(a[3], b[3]);
  float w = fmaxf(r - l, 0.f), h = fmaxf(bt - t, 0.f), s = w * h;
  float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return s / fmaxf(sa + sb - s, O_EPS);
}

/* Greedy NMS over boxes ALREADY sorted by descending score, as the bitmask kernel +
 * host reduce do (iou3d_kernel.cu:323-365, iou3d.cpp:117-164): box j is removed by an
 * earlier kept box i when iou(i,j) > thresh (strict).  mode 0: rotated BEV (N,5),
 * mode 1: 3-D (N,7, GPU iou_3d), mode 2: axis aligned (N,5). Returns number kept. */
int oracle_nms_sorted(const float *boxes, int N, float thresh, int mode, int64_t *keep) {
  int nk = 0;
  unsigned char *removed = (unsigned char *)__builtin_alloca(N > 0 ? N : 1);
  for (int i = 0; i < N; ++i) removed[i] = 0;
  for (int i = 0; i < N; ++i) {
    if (removed[i]) continue;
    keep[nk++] = i;
    for (int j = i + 1; j < N; ++j) {
      if (removed[j]) continue;
      float v;
      if (mode == 0) v = bev_iou5(boxes + 5 * i, boxes + 5 * j);
      else if (mode == 2) v = normal_iou5(boxes + 5 * i, boxes + 5 * j);
      else oracle_boxes_iou3d(boxes + 7 * i, 1, boxes + 7 * j, 1, &v, 1);
      if (v > thresh) removed[j] = 1;
    }
  }
  return nk;
}
