// Anchor target assignment of the SE-SSD training pipeline on gfx950 ("AssignTarget, nearest-IoU anchor
// matching over 70400 anchors"). Replaces the per-sample numpy / numba code that runs in the reference's DataLoader workers:
//   det3d/datasets/pipelines/preprocess.py:236-358 (AssignTarget) -> det3d/core/anchor/target_assigner.py:68-136 (assign_v2)
//   -> det3d/core/anchor/target_ops_v3.py:11-137 (create_target_np) with det3d/core/bbox/region_similarity.py:85-98
//   (NearestIouSimilarity: rbbox2d_to_near_bbox + iou_jit(eps=0), box_np_ops.py:354-366,1008-1046) and
//   box_np_ops.second_box_encode (:52-110).
// Two launches over the anchors (one thread per anchor, the <= 128 ground-truth boxes of the sample in LDS):
//   1. IoU of the anchor's nearest axis-aligned box with every ground-truth box in the same float32 operation order as
//      iou_jit; per-anchor max / first argmax; per-ground-truth max by wave reduction + atomicMax on the float bits
//      (IoU >= 0, so the unsigned order is the float order; a max is order independent)
//   2. labels: forced positives (anchors that attain a ground truth's maximum, recomputed bit-identically), positives at
//      IoU >= matched, background below unmatched (forced positives win), ignore (-1) in between; regression targets
//      (second_box_encode) and weights of the foreground anchors.
// HBM-bound and tiny: 70400 x 28 B read twice, 70400 x 40 B written.
#include "common.hpp"

namespace {

constexpr int MAX_GT = 128;
constexpr float PI_F = 3.14159274101257324f;      // np.float32(np.pi)
constexpr float PI_4_F = 0.785398185253143311f;   // np.float32(np.pi / 4)

struct NearBox {
  float x1, y1, x2, y2;
};

// box_np_ops.rbbox2d_to_near_bbox on [x, y, w, l, r]
__device__ __forceinline__ NearBox near_box(float x, float y, float w, float l, float r) {
  const float folded = fabsf(r - floorf(__fdiv_rn(r, PI_F) + 0.5f) * PI_F);
  const bool swap = folded > PI_4_F;
  const float ww = swap ? l : w, ll = swap ? w : l;
  NearBox b;
  b.x1 = x - ww / 2.f; b.y1 = y - ll / 2.f; b.x2 = x + ww / 2.f; b.y2 = y + ll / 2.f;
  return b;
}

// iou_jit(eps = 0): boxes = anchors, query = ground truth
__device__ __forceinline__ float near_iou(const NearBox& a, const NearBox& g, float area_g) {
  const float iw = fminf(a.x2, g.x2) - fmaxf(a.x1, g.x1);
  if (!(iw > 0.f)) return 0.f;
  const float ih = fminf(a.y2, g.y2) - fmaxf(a.y1, g.y1);
  if (!(ih > 0.f)) return 0.f;
  const float inter = iw * ih;
  const float ua = (a.x2 - a.x1) * (a.y2 - a.y1) + area_g - inter;
  return __fdiv_rn(inter, ua);
}

__device__ __forceinline__ void load_gt(const float* __restrict__ gt, int m, NearBox* sg, float* sarea) {
  for (int k = threadIdx.x; k < m; k += blockDim.x) {
    const float* g = gt + (size_t)k * 7;
    const NearBox b = near_box(g[0], g[1], g[3], g[4], g[6]);
    sg[k] = b;
    sarea[k] = (b.x2 - b.x1) * (b.y2 - b.y1);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void assign_iou_kernel(const float* __restrict__ anchors, int n, const float* __restrict__ gt,
                                                          int m, float* __restrict__ a_max, int* __restrict__ a_arg,
                                                          unsigned* __restrict__ gmax_bits) {
  __shared__ NearBox sg[MAX_GT];
  __shared__ float sarea[MAX_GT];
  load_gt(gt, m, sg, sarea);
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  NearBox a = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const float* p = anchors + (size_t)i * 7;
    a = near_box(p[0], p[1], p[3], p[4], p[6]);
  }
  float best = -1.f;
  int arg = 0;
  for (int k = 0; k < m; ++k) {
    const float v = live ? near_iou(a, sg[k], sarea[k]) : 0.f;
    if (v > best) { best = v; arg = k; }  // strict: first maximum, like numpy argmax
    float w = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w = fmaxf(w, __shfl_xor(w, o, 64));
    if ((threadIdx.x & 63) == 0 && w > 0.f) atomicMax(&gmax_bits[k], __float_as_uint(w));
  }
  if (live) { a_max[i] = best; a_arg[i] = arg; }
}

__global__ __launch_bounds__(256) void assign_label_kernel(const float* __restrict__ anchors, int n, const float* __restrict__ gt,
                                                            const int* __restrict__ gt_classes, int m, float matched,
                                                            float unmatched, const float* __restrict__ a_max,
                                                            const int* __restrict__ a_arg, const unsigned* __restrict__ gmax_bits,
                                                            int* __restrict__ labels, float* __restrict__ targets,
                                                            float* __restrict__ weights, int* __restrict__ gt_id) {
  __shared__ NearBox sg[MAX_GT];
  __shared__ float sarea[MAX_GT];
  load_gt(gt, m, sg, sarea);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = anchors + (size_t)i * 7;
  float* t = targets + (size_t)i * 7;
  if (m == 0) {  // target_ops_v3.py:90-91,99-100: everything is background
    labels[i] = 0; weights[i] = 0.f; gt_id[i] = -1;
#pragma unroll
    for (int q = 0; q < 7; ++q) t[q] = 0.f;
    return;
  }
  const NearBox a = near_box(p[0], p[1], p[3], p[4], p[6]);
  bool force = false;
  for (int k = 0; k < m; ++k) {
    const unsigned gb = gmax_bits[k];
    if (gb == 0u) continue;  // a ground truth that overlaps no anchor forces nothing (:65-67)
    force = force || (near_iou(a, sg[k], sarea[k]) == __uint_as_float(gb));
  }
  const float best = a_max[i];
  const int arg = a_arg[i];
  const int cls = gt_classes ? gt_classes[arg] : 1;
  int label = -1, gid = -1;
  if (force || best >= matched) { label = cls; gid = arg; }
  const bool fg = label > 0;
  if (best < unmatched) label = 0;
  if (force) label = cls;
  labels[i] = label;
  weights[i] = label > 0 ? 1.f : 0.f;
  gt_id[i] = fg ? gid : -1;
  if (fg) {  // box_np_ops.second_box_encode (:66-110), plain residual yaw
    const float* g = gt + (size_t)arg * 7;
    const float diag = sqrtf(p[4] * p[4] + p[3] * p[3]);
    t[0] = __fdiv_rn(g[0] - p[0], diag);
    t[1] = __fdiv_rn(g[1] - p[1], diag);
    t[2] = __fdiv_rn(g[2] - p[2], p[5]);
    t[3] = logf(__fdiv_rn(g[3], p[3]));
    t[4] = logf(__fdiv_rn(g[4], p[4]));
    t[5] = logf(__fdiv_rn(g[5], p[5]));
    t[6] = g[6] - p[6];
  } else {
#pragma unroll
    for (int q = 0; q < 7; ++q) t[q] = 0.f;
  }
}

}  // namespace

extern "C" {

size_t sessd_assign_targets_workspace_bytes(int num_anchors) {
  return sessd_align((size_t)MAX_GT * 4, 256) + 2 * sessd_align((size_t)num_anchors * 4, 256);
}

// anchors (n,7), gt_boxes (m,7) [x,y,z,w,l,h,r] float32, gt_classes (m,) int32 or NULL (all 1), m <= 128. Outputs (device):
// labels (n,) int32 in {-1 ignore, 0 background, class}, bbox_targets (n,7), bbox_outside_weights (n,), gt_id (n,) = index of
// the assigned ground truth for the foreground anchors, -1 elsewhere (the reference's positive_gt_id is gt_id[gt_id >= 0]).
int sessd_assign_targets(const float* anchors, int num_anchors, const float* gt_boxes, const int* gt_classes, int num_gt,
                         float matched_threshold, float unmatched_threshold, int* labels, float* bbox_targets,
                         float* bbox_outside_weights, int* gt_id, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (num_anchors <= 0 || num_gt < 0 || num_gt > MAX_GT) return SESSD_EINVAL;
  if (workspace_bytes < sessd_assign_targets_workspace_bytes(num_anchors)) return SESSD_EWORKSPACE;
  char* base = (char*)workspace;
  unsigned* gmax = (unsigned*)base;
  float* a_max = (float*)(base + sessd_align((size_t)MAX_GT * 4, 256));
  int* a_arg = (int*)((char*)a_max + sessd_align((size_t)num_anchors * 4, 256));
  SESSD_FILL(gmax, 0u, MAX_GT, stream);
  const int blocks = sessd_divup(num_anchors, 256);
  if (num_gt > 0) {
    SESSD_LAUNCH(assign_iou_kernel, dim3(blocks), dim3(256), 0, stream, anchors, num_anchors, gt_boxes, num_gt, a_max, a_arg, gmax);
    SESSD_CHECK_LAUNCH();
  }
  SESSD_LAUNCH(assign_label_kernel, dim3(blocks), dim3(256), 0, stream, anchors, num_anchors, gt_boxes, gt_classes, num_gt,
               matched_threshold, unmatched_threshold, a_max, a_arg, gmax, labels, bbox_targets, bbox_outside_weights, gt_id);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
