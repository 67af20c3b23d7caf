// HOST-side box geometry of the training data path: the box-level decisions of GT-AUG sampling and per-object
// noise -- det3d/core/sampler/preprocess.py:944-1027 box_collision_test and :579-611 noise_per_box (numba kernels in the
// reference). They are sequential by nature (a box's accepted move is what the next boxes are tested against) and tiny (tens of
// boxes), so they stay on the host even when the point cloud lives on the device; as vectorised numpy they were the largest
// part of the device-mode stage (17 collision-test calls with ~40 small array operations each per sample). Plain C++ here,
// same arithmetic in the same order and precision as the numpy mirror (float32 or float64 by the arrays' dtype, every product
// and difference rounded separately: this file is compiled with -ffp-contract=off), so decisions are identical
// (tests/test_datapath_cpu.py: the reference's own goldens; tests/test_host_boxes_cpu.py: random quads vs the numpy form).
// No device code in this file.
#include <stdint.h>
#include <algorithm>
#include "common.hpp"

namespace {

template <typename T>
inline bool ccw(const T* P, const T* Q, const T* R) {
  return (R[1] - P[1]) * (Q[0] - P[0]) > (Q[1] - P[1]) * (R[0] - P[0]);
}

// every corner of `inner` strictly inside `outer` (edges k -> k+1; clockwise flips the edge vectors)
template <typename T>
inline bool quad_inside(const T* outer, const T* inner, bool clockwise) {
  for (int k = 0; k < 4; ++k) {
    const T* o0 = outer + 2 * k;
    const T* o1 = outer + 2 * ((k + 1) & 3);
    T vx = o0[0] - o1[0], vy = o0[1] - o1[1];
    if (clockwise) { vx = -vx; vy = -vy; }
    for (int m = 0; m < 4; ++m) {
      const T cross = vy * (o0[0] - inner[2 * m]) - vx * (o0[1] - inner[2 * m + 1]);
      if (cross >= 0) return false;
    }
  }
  return true;
}

template <typename T>
bool quads_collide(const T* a, const T* b, bool clockwise) {
  T amin[2], amax[2], bmin[2], bmax[2];
  for (int d = 0; d < 2; ++d) {
    amin[d] = amax[d] = a[d];
    bmin[d] = bmax[d] = b[d];
    for (int k = 1; k < 4; ++k) {
      amin[d] = std::min(amin[d], a[2 * k + d]); amax[d] = std::max(amax[d], a[2 * k + d]);
      bmin[d] = std::min(bmin[d], b[2 * k + d]); bmax[d] = std::max(bmax[d], b[2 * k + d]);
    }
  }
  const T iw = std::min(amax[0], bmax[0]) - std::max(amin[0], bmin[0]);
  const T ih = std::min(amax[1], bmax[1]) - std::max(amin[1], bmin[1]);
  if (!(iw > 0 && ih > 0)) return false;
  for (int k = 0; k < 4; ++k) {
    const T* A = a + 2 * k;
    const T* B = a + 2 * ((k + 1) & 3);
    for (int l = 0; l < 4; ++l) {
      const T* C = b + 2 * l;
      const T* D = b + 2 * ((l + 1) & 3);
      if ((ccw(A, C, D) != ccw(B, C, D)) && (ccw(A, B, C) != ccw(A, B, D))) return true;
    }
  }
  return quad_inside(a, b, clockwise) || quad_inside(b, a, clockwise);
}

template <typename T>
void collision_matrix(const T* boxes, int n, const T* qboxes, int k, bool clockwise, uint8_t* out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < k; ++j) out[(size_t)i * k + j] = quads_collide(boxes + 8 * i, qboxes + 8 * j, clockwise) ? 1 : 0;
}

template <typename T>
void noise_per_box(T* corners, const T* centers, const uint8_t* valid, const double* loc_xy, const double* sin_r, const double* cos_r,
                   int n, int tries, int64_t* chosen) {
  for (int i = 0; i < n; ++i) {
    chosen[i] = -1;
    if (!valid[i]) continue;
    T local[8];
    for (int k = 0; k < 4; ++k) {
      local[2 * k] = corners[8 * i + 2 * k] - centers[2 * i];
      local[2 * k + 1] = corners[8 * i + 2 * k + 1] - centers[2 * i + 1];
    }
    for (int t = 0; t < tries; ++t) {
      const size_t it = (size_t)i * tries + t;
      const double s = sin_r[it], c = cos_r[it];
      const double sx = (double)centers[2 * i] + loc_xy[2 * it], sy = (double)centers[2 * i + 1] + loc_xy[2 * it + 1];
      T cand[8];
      for (int k = 0; k < 4; ++k) {
        // numpy: the rotated corner is formed in float64, stored in the boxes' dtype, then shifted in float64 and stored again
        const T rx = (T)((double)local[2 * k] * c + (double)local[2 * k + 1] * s);
        const T ry = (T)((double)local[2 * k] * -s + (double)local[2 * k + 1] * c);
        cand[2 * k] = (T)((double)rx + sx);
        cand[2 * k + 1] = (T)((double)ry + sy);
      }
      bool hit = false;
      for (int j = 0; j < n && !hit; ++j)
        if (j != i) hit = quads_collide(cand, corners + 8 * j, true);
      if (!hit) {
        chosen[i] = t;
        for (int e = 0; e < 8; ++e) corners[8 * i + e] = cand[e];
        break;
      }
    }
  }
}

}  // namespace

extern "C" {

// out[i][j] = 1 if BEV quadrilateral boxes[i] (4 corners x (x, y)) collides with qboxes[j]: bounding rectangles overlap and two
// edges cross or one lies inside the other (preprocess.py:944-1027). is_f32: the arrays' element type (float32 / float64).
int sessd_box_collision_host(const void* boxes, int n, const void* qboxes, int k, int is_f32, int clockwise, uint8_t* out) {
  if (n < 0 || k < 0 || (n && !boxes) || (k && !qboxes) || (n && k && !out)) return SESSD_EINVAL;
  if (is_f32)
    collision_matrix((const float*)boxes, n, (const float*)qboxes, k, clockwise != 0, out);
  else
    collision_matrix((const double*)boxes, n, (const double*)qboxes, k, clockwise != 0, out);
  return SESSD_OK;
}

// preprocess.py:579-611 noise_per_box: corners (n, 4, 2) of the boxes' BEV footprints (updated in place with the accepted
// moves), centers (n, 2), valid (n), candidate moves loc_xy (n, tries, 2) float64 and the sines / cosines of the candidate
// rotations (n, tries) float64 -> chosen[i] = index of the first candidate of box i whose moved footprint collides with no
// other box in its current place, -1 if none (or box not valid). Boxes are processed in order.
int sessd_noise_per_box_host(void* corners, const void* centers, const uint8_t* valid, const double* loc_xy, const double* sin_r,
                             const double* cos_r, int n, int tries, int is_f32, int64_t* chosen) {
  if (n < 0 || tries < 0) return SESSD_EINVAL;
  if (n == 0) return SESSD_OK;
  if (!corners || !centers || !valid || !chosen || (tries && (!loc_xy || !sin_r || !cos_r))) return SESSD_EINVAL;
  if (is_f32)
    noise_per_box((float*)corners, (const float*)centers, valid, loc_xy, sin_r, cos_r, n, tries, chosen);
  else
    noise_per_box((double*)corners, (const double*)centers, valid, loc_xy, sin_r, cos_r, n, tries, chosen);
  return SESSD_OK;
}

}  // extern "C"
