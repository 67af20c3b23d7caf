/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Plain-C restatement of DI-NMS, the IoU-weighted rotated NMS of
 *   det3d/ops/nms/nms_cpu.h:173-384   IOU_weighted_rotate_non_max_suppression_cpu<float>
 * as called by det3d/ops/nms/nms_cpu.py:52-93 (rotate_weighted_nms_cc) from
 * det3d/core/bbox/box_torch_ops.py:552-621 (rotate_weighted_nms).
 *
 * Repeat: take the not-yet-suppressed box A with the largest ORIGINAL score (first on ties), mark it; over ALL boxes j
 * (suppressed or not, A itself included) with a non-empty polygon intersection and overlap = |A n j| / |A u j|:
 *   same label and overlap > 0                  -> cnt += overlap * iou_pred[j]
 *   same label and overlap > suppressed_thresh  -> score_box = max(score_box, normalised score j); weighted box sums with
 *                                                  w = exp(-(1 - overlap)^2 / sigma^2(distance of A from the origin)) * iou_pred[j]
 *   j not suppressed, stand-up IoU(A, j) > 0, overlap >= suppressed_thresh -> suppress j (remembered)
 * cnt > cnt_thresh: A is kept with the weighted average box, score_box * (normalisation maximum), its label and direction;
 * otherwise the boxes suppressed in this pass come back (A itself does not).
 * Scores are first damped by centerness (optional) and divided by their maximum.
 *
 * Arithmetic follows the float instantiation the Python wrapper reaches (float32 arrays): accumulators are float, the
 * pow / sqrt / exp calls take their double overloads and round back to float on assignment. The reference intersects
 * and unites the quads with boost::geometry, which is not installed here; the oracle uses the double-precision clipper of
 * rotate_nms.c and |A u B| = |A| + |B| - |A n B|, rounded to float like the reference's assignments.
 * Pinned by: the reference's own nms_cpu.h COMPILED FROM SOURCE with a boost::geometry stand-in (oracle/boost_shim, the same
 * clipping arithmetic; oracle/build.py build_ref_nms) -> tests/golden/nms_cpu_ref.npz, tests/test_nms_cpu_ref_cpu.py: identical
 * selection, order, labels and directions, boxes and scores to 2e-5 on 16 clustered cases. That pins the CONTROL FLOW of
 * :173-384; boost's own polygon-area arithmetic remains PARITY UNPINNED. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

double oracle_quad_intersection_area(const float *P, const float *Q);

static double quad_abs_area(const float *P) {
  double a = 0;
  for (int i = 0; i < 4; ++i) {
    int j = (i + 1) & 3;
    a += (double)P[2 * i] * P[2 * j + 1] - (double)P[2 * j] * P[2 * i + 1];
  }
  return fabs(0.5 * a);
}

/* boxes (N,7), corners (N,4,2), standup_iou (N,N), scores / iou_preds (N), labels / dirs (N) int32, anchors (N, anchor_stride) or
 * NULL when centerness_c == 0. Outputs have capacity N; returns the number of boxes kept. */
int oracle_di_nms(const float *boxes, const float *corners, const float *standup_iou, int N, const float *scores,
                  const float *iou_preds, const int32_t *labels, const int32_t *dirs, const float *anchors, int anchor_stride,
                  float cnt_thresh, const float *sigma_dist_interval, int n_interval, const float *sigma_square,
                  float suppressed_thresh, int centerness_c, float *boxes_ret, float *scores_ret, int32_t *labels_ret,
                  int32_t *dirs_ret, int32_t *keep) {
  if (N <= 0) return 0;
  int *suppressed = (int *)calloc((size_t)N, sizeof(int));
  int *recover = (int *)malloc((size_t)N * sizeof(int));
  float *scores_rw = (float *)malloc((size_t)N * sizeof(float));
  for (int i = 0; i < N; ++i) scores_rw[i] = scores[i];
  if (centerness_c == 1) {
    float *cen = (float *)malloc((size_t)N * sizeof(float));
    float sum = 0;
    for (int i = 0; i < N; ++i) {
      float dist = (float)sqrt(pow(boxes[i * 7] - anchors[(size_t)i * anchor_stride], 2) +
                               pow(boxes[i * 7 + 1] - anchors[(size_t)i * anchor_stride + 1], 2));
      cen[i] = (float)exp(dist);
    }
    for (int i = 0; i < N; ++i) sum += cen[i];
    for (int i = 0; i < N; ++i) {
      cen[i] /= sum;
      scores_rw[i] *= (1 - cen[i]);
    }
    free(cen);
  }
  float score_max4norm = -10000;
  for (int i = 0; i < N; ++i)
    if (scores_rw[i] > score_max4norm) score_max4norm = scores_rw[i];
  for (int i = 0; i < N; ++i) scores_rw[i] /= score_max4norm;
  int nkeep = 0;
  for (;;) {
    float score_max = -1;
    int idx_max = -1, all_checked = 1;
    for (int i = 0; i < N; ++i) {
      if (suppressed[i] == 1) continue;
      all_checked = 0;
      if (scores[i] > score_max) { score_max = scores[i]; idx_max = i; }
    }
    if (all_checked) break;
    if (idx_max < 0) break; /* every remaining score <= -1: the reference would index out of range here */
    float dist2origin = (float)sqrt(pow(boxes[idx_max * 7], 2) + pow(boxes[idx_max * 7 + 1], 2));
    suppressed[idx_max] = 1;
    float weight_pos[7] = {0, 0, 0, 0, 0, 0, 0}, avg_pos[7] = {0, 0, 0, 0, 0, 0, 0};
    float score_box = -1, cnt = 0;
    int nrec = 0;
    const float *A = corners + (size_t)idx_max * 8;
    const double area_a = quad_abs_area(A);
    for (int j = 0; j < N; ++j) {
      const float *B = corners + (size_t)j * 8;
      const double inter_d = oracle_quad_intersection_area(A, B);
      if (!(inter_d > 0)) continue;                 /* poly_inter empty */
      const float inter_area = (float)inter_d;
      const float union_area = (float)(area_a + quad_abs_area(B) - inter_d);
      if (!(union_area > 0)) continue;              /* poly_union empty */
      const float overlap = inter_area / union_area;
      const int same = labels[j] == labels[idx_max];
      if (overlap > 0 && same) cnt += overlap * iou_preds[j];
      if (overlap > suppressed_thresh && same) {
        if (score_box < scores_rw[j]) score_box = scores_rw[j];
        float iou_weight = 0;
        for (int k = 0; k + 1 < n_interval; ++k)
          if (dist2origin >= sigma_dist_interval[k] && dist2origin < sigma_dist_interval[k + 1])
            iou_weight = (float)exp(-pow(1 - overlap, 2) / sigma_square[k]);
        for (int k = 0; k < 7; ++k) {
          avg_pos[k] += iou_weight * iou_preds[j] * boxes[j * 7 + k];
          weight_pos[k] += iou_weight * iou_preds[j];
        }
      }
      if (suppressed[j] != 1 && standup_iou[(size_t)idx_max * N + j] > 0 && overlap >= suppressed_thresh) {
        suppressed[j] = 1;
        recover[nrec++] = j;
      }
    }
    if (cnt > cnt_thresh) {
      keep[nkeep] = idx_max;
      scores_ret[nkeep] = score_box * score_max4norm;
      for (int k = 0; k < 7; ++k) boxes_ret[nkeep * 7 + k] = avg_pos[k] / weight_pos[k];
      labels_ret[nkeep] = labels[idx_max];
      dirs_ret[nkeep] = dirs[idx_max];
      ++nkeep;
    } else {
      for (int k = 0; k < nrec; ++k) suppressed[recover[k]] = 0;
    }
  }
  free(suppressed); free(recover); free(scores_rw);
  return nkeep;
}
