/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Plain-C restatement of the reference's numba-CUDA rotated IoU (the [cx,cy,w,l,angle] convention used by
 * rotate_iou_gpu / rotate_iou_gpu_eval / rotate_nms_gpu and therefore by the KITTI AP evaluation):
 *   det3d/ops/nms/nms_gpu.py:183-199  trangle_area, area (fan of |triangle| areas from vertex 0)
 *   det3d/ops/nms/nms_gpu.py:202-239  sort_vertex_in_convex_polygon (pseudo-angle key, insertion sort)
 *   det3d/ops/nms/nms_gpu.py:242-286  line_segment_intersection
 *   det3d/ops/nms/nms_gpu.py:325-368  point_in_quadrilateral, quadrilateral_intersection
 *   det3d/ops/nms/nms_gpu.py:371-419  rbbox_to_corners, inter, devRotateIoU
 *   det3d/ops/nms/nms_gpu.py:580-633  devRotateIoUEval, rotate_iou_kernel_eval (note: query box FIRST)
 * Pinned by tests/golden/rotate_iou_numba_ref.npz (the numba device functions executed as Python).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static float tri_area(const float *a, const float *b, const float *c) {
  return ((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f;
}

static float fan_area(const float *p, int n) {
  float s = 0.0f;
  for (int i = 0; i < n - 2; ++i) s += fabsf(tri_area(p, p + 2 * i + 2, p + 2 * i + 4));
  return s;
}

static void sort_vertices(float *p, int n) {
  if (n <= 0) return;
  float cx = 0, cy = 0, vs[16];
  for (int i = 0; i < n; ++i) { cx += p[2 * i]; cy += p[2 * i + 1]; }
  cx /= n; cy /= n;
  for (int i = 0; i < n; ++i) {
    float vx = p[2 * i] - cx, vy = p[2 * i + 1] - cy;
    float d = sqrtf(vx * vx + vy * vy);
    vx = vx / d; vy = vy / d;
    if (vy < 0) vx = -2 - vx;
    vs[i] = vx;
  }
  for (int i = 1; i < n; ++i) {
    if (vs[i - 1] > vs[i]) {
      float temp = vs[i], tx = p[2 * i], ty = p[2 * i + 1];
      int j = i;
      while (j > 0 && vs[j - 1] > temp) {
        vs[j] = vs[j - 1]; p[2 * j] = p[2 * j - 2]; p[2 * j + 1] = p[2 * j - 1];
        --j;
      }
      vs[j] = temp; p[2 * j] = tx; p[2 * j + 1] = ty;
    }
  }
}

static int seg_inter(const float *p1, const float *p2, int i, int j, float *out) {
  float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
  float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
  float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
  int acd = DA1 * CA0 > CA1 * DA0;
  int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
  if (acd != bcd) {
    int abc = CA1 * BA0 > BA1 * CA0;
    int abd = DA1 * BA0 > BA1 * DA0;
    if (abc != abd) {
      float DC0 = D0 - C0, DC1 = D1 - C1;
      float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
      float DH = BA1 * DC0 - BA0 * DC1;
      out[0] = (ABBA * DC0 - BA0 * CDDC) / DH;
      out[1] = (ABBA * DC1 - BA1 * CDDC) / DH;
      return 1;
    }
  }
  return 0;
}

static int in_quad(float x, float y, const float *c) {
  float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
  float ap0 = x - c[0], ap1 = y - c[1];
  float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

static void corners_of(const float *r, float *c) {
  float a_cos = cosf(r[4]), a_sin = sinf(r[4]);
  float xs[4] = {-r[2] / 2, -r[2] / 2, r[2] / 2, r[2] / 2}, ys[4] = {-r[3] / 2, r[3] / 2, r[3] / 2, -r[3] / 2};
  for (int i = 0; i < 4; ++i) {
    c[2 * i] = a_cos * xs[i] + a_sin * ys[i] + r[0];
    c[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + r[1];
  }
}

float oracle_rotate_inter(const float *r1, const float *r2) {
  float c1[8], c2[8], pts[16 + 16], tmp[2];
  corners_of(r1, c1);
  corners_of(r2, c2);
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    if (in_quad(c1[2 * i], c1[2 * i + 1], c2)) { pts[2 * n] = c1[2 * i]; pts[2 * n + 1] = c1[2 * i + 1]; ++n; }
    if (in_quad(c2[2 * i], c2[2 * i + 1], c1)) { pts[2 * n] = c2[2 * i]; pts[2 * n + 1] = c2[2 * i + 1]; ++n; }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_inter(c1, c2, i, j, tmp) && n < 16) { pts[2 * n] = tmp[0]; pts[2 * n + 1] = tmp[1]; ++n; }
  if (n > 8) n = 8; /* the reference's local array holds 8 points (16 floats) */
  sort_vertices(pts, n);
  return fan_area(pts, n);
}

/* criterion: -1 IoU | 0 inter/area(first) | 1 inter/area(second) | else inter  (first, second as PASSED) */
float oracle_rotate_iou_pair(const float *r1, const float *r2, int criterion) {
  float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
  float it = oracle_rotate_inter(r1, r2);
  if (criterion == -1) return it / (a1 + a2 - it);
  if (criterion == 0) return it / a1;
  if (criterion == 1) return it / a2;
  return it;
}

/* rotate_iou_gpu_eval: out[n][k] = devRotateIoUEval(query[k], boxes[n], criterion)  (nms_gpu.py:626-631) */
void oracle_rotate_iou_eval(const float *boxes, int N, const float *query, int K, int criterion, float *out) {
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) out[(size_t)n * K + k] = oracle_rotate_iou_pair(query + 5 * k, boxes + 5 * n, criterion);
}
