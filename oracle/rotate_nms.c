/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Plain-C restatement of the rotated NMS that MultiGroupHead.predict really calls:
 *   det3d/core/bbox/box_torch_ops.py:527-548   rotate_nms (topk, D2H, keep[:post], H2D)
 *   det3d/ops/nms/nms_cpu.py:40-51             rotate_nms_cc (order, corners, stand-up boxes, iou_jit eps=0)
 *   det3d/core/bbox/box_np_ops.py:267-294,433-446,512-532  corners_nd / rotation_2d / center_to_corner_box2d
 *   det3d/core/bbox/box_np_ops.py:1007-1045    iou_jit
 *   det3d/ops/nms/nms_cpu.h:72-168             rotate_non_max_suppression_cpu (greedy loop)
 * The reference intersects the two quads with boost::geometry (absent here, so this
 * piece cannot be compiled from the reference sources); the oracle clips convex quads
 * with Sutherland-Hodgman in double precision and uses area(P u Q) = |P| + |Q| - |P n Q|.
 * Corner, stand-up box and prefilter arithmetic is float32 exactly as numpy does it.
 * Pinned by: tests/golden/nms_helpers.npz (reference numpy helpers run from source),
 * cross-agreement of the polygon IoU with the compiled iou3d reference (oracle/_ref), and the
 * greedy loop itself by the reference's own nms_cpu.h compiled from source with a boost::geometry
 * stand-in (oracle/boost_shim; tests/golden/nms_cpu_ref.npz, tests/test_nms_cpu_ref_cpu.py): the
 * control flow of :72-168 is pinned, boost's polygon-area arithmetic is not.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* dets row: x, y, w, l, r ; corners out: (4,2) float32.
 * corners_nd(origin 0.5): unit offsets (-.5,-.5),(-.5,.5),(.5,.5),(.5,-.5) times (w,l);
 * rotation_2d: [x', y'] = [x cos + y sin, -x sin + y cos]  (einsum with rot_mat_T = [[c,-s],[s,c]]) */
void oracle_box2d_corners(const float *det, float *corners) {
  static const float ux[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, uy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
  float s = sinf(det[4]), c = cosf(det[4]);
  for (int k = 0; k < 4; ++k) {
    volatile float px = det[2] * ux[k], py = det[3] * uy[k];
    /* einsum "aij,jka->aik": x' = px*c + py*s ; y' = px*(-s) + py*c, float32 products and sums */
    volatile float a = px * c, b = py * s, d = px * (-s), e = py * c;
    volatile float rx = a + b, ry = d + e;
    corners[2 * k] = rx + det[0];
    corners[2 * k + 1] = ry + det[1];
  }
}

static double quad_area(const double *p, int n) {
  double a = 0;
  for (int i = 0; i < n; ++i) {
    int j = (i + 1) % n;
    a += p[2 * i] * p[2 * j + 1] - p[2 * j] * p[2 * i + 1];
  }
  return 0.5 * a;
}

/* area of intersection of two convex quads (any orientation), double precision */
double oracle_quad_intersection_area(const float *P, const float *Q) {
  double subj[32], tmp[32], clip[8];
  int ns = 4;
  for (int i = 0; i < 8; ++i) { subj[i] = P[i]; clip[i] = Q[i]; }
  double ca = quad_area(clip, 4);
  if (ca == 0) return 0;
  double sgn = ca > 0 ? 1.0 : -1.0;
  for (int e = 0; e < 4 && ns > 0; ++e) {
    double ax = clip[2 * e], ay = clip[2 * e + 1], bx = clip[2 * ((e + 1) % 4)], by = clip[2 * ((e + 1) % 4) + 1];
    int nt = 0;
    for (int i = 0; i < ns; ++i) {
      int j = (i + 1) % ns;
      double cx = subj[2 * i], cy = subj[2 * i + 1], dx = subj[2 * j], dy = subj[2 * j + 1];
      double sc = sgn * ((bx - ax) * (cy - ay) - (by - ay) * (cx - ax));
      double sd = sgn * ((bx - ax) * (dy - ay) - (by - ay) * (dx - ax));
      if (sc >= 0) { tmp[2 * nt] = cx; tmp[2 * nt + 1] = cy; nt++; }
      if ((sc >= 0) != (sd >= 0)) {
        double t = sc / (sc - sd);
        tmp[2 * nt] = cx + t * (dx - cx);
        tmp[2 * nt + 1] = cy + t * (dy - cy);
        nt++;
      }
    }
    ns = nt;
    for (int i = 0; i < 2 * ns; ++i) subj[i] = tmp[i];
  }
  if (ns < 3) return 0;
  return fabs(quad_area(subj, ns));
}

/* polygon IoU of two (4,2) float32 corner sets, as nms_cpu.h:139-157 computes it */
double oracle_quad_iou(const float *P, const float *Q) {
  double inter = oracle_quad_intersection_area(P, Q);
  double p[8], q[8];
  for (int i = 0; i < 8; ++i) { p[i] = P[i]; q[i] = Q[i]; }
  double uni = fabs(quad_area(p, 4)) + fabs(quad_area(q, 4)) - inter;
  return uni > 0 ? inter / uni : 0.0;
}

/* dets (K,6) float32 rows [x,y,w,l,r,score]; `order` = indices by descending score
 * (ties: the caller decides; the reference's numpy quicksort leaves ties unspecified).
 * keep: out, capacity K. Returns number kept (ALL kept; the caller truncates to post_max_size
 * like box_torch_ops.py:543). near_thresh (optional, may be NULL) counts candidate pairs
 * whose IoU lies within `margin` of the threshold (the tests use it to flag inputs on
 * which a float32 implementation may legitimately disagree). */
static int oracle_rotate_nms_impl(const float *dets, int K, const int32_t *order, float thresh, int32_t *keep, double margin,
                                 int32_t *near_thresh, int32_t *near_pairs, int near_cap, const int32_t *forced, int n_forced) {
  float *corners = (float *)malloc((size_t)K * 8 * sizeof(float));
  float *standup = (float *)malloc((size_t)K * 4 * sizeof(float));
  unsigned char *sup = (unsigned char *)calloc(K > 0 ? K : 1, 1);
  for (int i = 0; i < K; ++i) {
    oracle_box2d_corners(dets + 6 * i, corners + 8 * i);
    float x0 = corners[8 * i], y0 = corners[8 * i + 1], x1 = x0, y1 = y0;
    for (int k = 1; k < 4; ++k) { /* corner_to_standup_nd: min / max over corners */
      float x = corners[8 * i + 2 * k], y = corners[8 * i + 2 * k + 1];
      x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1;
      y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
    }
    standup[4 * i] = x0; standup[4 * i + 1] = y0; standup[4 * i + 2] = x1; standup[4 * i + 3] = y1;
  }
  int nk = 0, nnear = 0;
  for (int _i = 0; _i < K; ++_i) {
    int i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    for (int _j = _i + 1; _j < K; ++_j) {
      int j = order[_j];
      if (sup[j]) continue;
      /* iou_jit(eps=0): positive only when both overlaps are strictly positive (float32) */
      const float *a = standup + 4 * i, *b = standup + 4 * j;
      float iw = (a[2] < b[2] ? a[2] : b[2]) - (a[0] > b[0] ? a[0] : b[0]);
      if (!(iw > 0)) continue;
      float ih = (a[3] < b[3] ? a[3] : b[3]) - (a[1] > b[1] ? a[1] : b[1]);
      if (!(ih > 0)) continue;
      volatile float ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - iw * ih;
      volatile float siou = iw * ih / ua;
      if (siou <= 0.0f) continue;
      /* nms_cpu.h:139: nothing happens when boost returns an empty intersection */
      if (oracle_quad_intersection_area(corners + 8 * i, corners + 8 * j) <= 0) continue;
      double ov = oracle_quad_iou(corners + 8 * i, corners + 8 * j);
      if (fabs(ov - (double)thresh) < margin) {
        if (near_pairs && nnear < near_cap) { near_pairs[2 * nnear] = i; near_pairs[2 * nnear + 1] = j; }
        nnear++;
      }
      int suppress = ov >= (double)thresh;
      for (int f = 0; f < n_forced; ++f) /* decisions taken the other way on purpose (oracle/compare.py) */
        if (forced[3 * f] == i && forced[3 * f + 1] == j) suppress = forced[3 * f + 2];
      if (suppress) sup[j] = 1;
    }
  }
  if (near_thresh) *near_thresh = nnear;
  free(corners); free(standup); free(sup);
  return nk;
}

int oracle_rotate_nms(const float *dets, int K, const int32_t *order, float thresh, int32_t *keep, double margin,
                      int32_t *near_thresh) {
  return oracle_rotate_nms_impl(dets, K, order, thresh, keep, margin, near_thresh, NULL, 0, NULL, 0);
}

/* same, also listing the (kept i, candidate j) pairs whose IoU lies within `margin` of the threshold (first near_cap of them):
 * the decisions a float32 implementation may legitimately take the other way */
int oracle_rotate_nms_pairs(const float *dets, int K, const int32_t *order, float thresh, int32_t *keep, double margin,
                            int32_t *near_thresh, int32_t *near_pairs, int near_cap) {
  return oracle_rotate_nms_impl(dets, K, order, thresh, keep, margin, near_thresh, near_pairs, near_cap, NULL, 0);
}

/* same with `n_forced` decisions imposed: forced = (i, j, suppress) triples for pairs (kept i, candidate j) */
int oracle_rotate_nms_forced(const float *dets, int K, const int32_t *order, float thresh, int32_t *keep, double margin,
                             int32_t *near_thresh, int32_t *near_pairs, int near_cap, const int32_t *forced, int n_forced) {
  return oracle_rotate_nms_impl(dets, K, order, thresh, keep, margin, near_thresh, near_pairs, near_cap, forced, n_forced);
}
