// Orientation-aware distance-IoU loss of SE-SSD as ONE differentiable device op.
// Replaces det3d/models/losses/odious.py:837-900 (odiou_3D) and what it composes -- rbbox_to_corners :448-487,
// compute_vertex :15-276, sort_vertex :278-342, area_polygon :345-445, find_convex_hull / mbr_diag_convex_hull /
// mbr_diag_compute :506-643 -- i.e. per-box Python loops over numpy arrays on the HOST (a device->host->device round trip
// and a scipy ConvexHull call per box pair in every training iteration) with hand-written Jacobians.
//
// One thread per (target box g, predicted box q) pair evaluates
//   term = 1 - IoU3D(g, q) + |c_g - c_q|^2 / (diag_bev^2 + h_inter^2 + 1e-7) + 1.25 (1 - |cos(r_q - r_g)|)
// in float64 FORWARD-MODE automatic differentiation: every quantity carries its 7 partial derivatives with respect to
// q = [x, y, z, w, l, h, r], so the same pass yields d term / d q (the reference back-propagates to the prediction only;
// targets carry no gradient). Piecewise choices (which corners are inside, which edges cross, vertex order, hull, the
// minimum-area edge, min / max selections, clamps) are taken on the VALUES, exactly as the reference's subgradients do.
// diag_bev: diagonal of the minimum-area rectangle aligned with an edge of the convex hull of the 8 BEV corners, with the
// reference's angle folding (|fmod(atan2, pi/2)|, pi written 3.1415926). The hull here is Andrew's monotone chain and
// every hull edge is a candidate; the reference takes scipy's (Qhull's) vertex order and skips the closing edge, which
// makes its choice between equal-area candidates depend on Qhull's start vertex -- see oracle/odiou.py and DESIGN.md.
#include "odiou_core.hpp"

namespace {

// eight lanes per pair: lane c < 7 carries d / d q[c], all eight compute the value (lane 7 repeats component 6 and writes nothing)
__global__ __launch_bounds__(64) void odiou_kernel(const float* __restrict__ gboxes, const float* __restrict__ qboxes, int n,
                                                    float* __restrict__ term_out, float* __restrict__ grad_out) {
  const int i = blockIdx.x * 8 + (threadIdx.x >> 3), c = threadIdx.x & 7;
  if (i >= n) return;
  double g[NB], qv[NB], term, grad;
#pragma unroll
  for (int k = 0; k < NB; ++k) { g[k] = (double)gboxes[(size_t)i * NB + k]; qv[k] = (double)qboxes[(size_t)i * NB + k]; }
  odiou_eval(g, qv, c < NB ? c : NB - 1, &term, &grad);
  if (c == 0) term_out[i] = (float)term;
  if (c < NB) grad_out[(size_t)i * NB + c] = (float)grad;
}

}  // namespace

extern "C" {

// gboxes, qboxes (n, 7) float32 [x, y, z, w, l, h, r] (device) -> term (n,) and d term / d qboxes (n, 7), float32.
// The loss of odious.py:895-899 is 2 * sum_i(weights_i * term_i) / batch_size; its gradient 2 * weights_i / batch_size * grad_i.
int sessd_odiou3d(const float* gboxes, const float* qboxes, int n, float* term, float* grad_q, hipStream_t stream) {
  if (n < 0) return SESSD_EINVAL;
  if (n == 0) return SESSD_OK;
  SESSD_LAUNCH(odiou_kernel, dim3(sessd_divup(n, 8)), dim3(64), 0, stream, gboxes, qboxes, n, term, grad_q);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
