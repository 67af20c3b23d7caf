// SE-SSD training loss as CAPACITY-FORM device kernels: no boolean-mask indexing, no host-read shapes, nothing that
// synchronises -- so the whole loss (values AND the gradient with respect to the four head outputs) can sit inside the
// captured training iteration (sessd_hip.train.TrainStep.capture). SURVEY 8f row 1, BASELINE configs[2].
//
// Replaces, for the single-task car head of examples/second/configs/config.py:
//   det3d/models/bbox_heads/mg_head_sessd.py:706-808   MultiGroupHead.loss (focal + ODIoU + direction + IoU-prediction;
//                                                      smooth-L1 localisation for the log)
//   det3d/models/bbox_heads/mg_head_sessd.py:810-890   get_model_ema_loss (the teacher's own terms, log only)
//   det3d/models/bbox_heads/mg_head_sessd.py:618-704   consistency_loss, :573-607 nn_distance (return mode '10'),
//                                                      :525-571 prepare_loss_weights (NormByNumPositives), helpers :27-77
//   det3d/models/losses/losses.py:146-203 (WeightedSmoothL1Loss), :364-418 (SigmoidFocalLoss), :489-531 (softmax CE)
//   det3d/models/losses/odious.py:837-900 (odiou_3D)   through csrc/odiou_core.hpp
//   det3d/core/iou3d/iou3d_utils.py:32-52,197-252      (boxes_iou_bev_gpu, boxes_aligned_iou3d_gpu) through csrc/geom.hpp
//   det3d/torchie/trainer/trainer_sessd.py:267         loss += consistency_loss * consistency_weight
// The reference builds these from ~150 torch ops with boolean-mask gathers (`box[pos]`), `if mask.sum() > 0` branches and a
// Python loop over the samples of the batch; every one of those reads a shape back to the host.
//
// Launches (all on one stream; counts stay on the device):
//   1 hl_count    per (256-anchor block, sample, network): positives (label > 0) and consistency candidates
//                 (sigmoid(cls) >= 0.3 and decoded centre inside post_center_range)
//   2 hl_anchor   per anchor: focal loss + its gradient, direction softmax loss + gradient and the logged smooth-L1 terms on the
//                 positives; ORDERED compaction (block prefix from the counts of launch 1) of the positives and of the
//                 consistency candidates (decoded; the teacher's mapped into the student frame: flip, rotation, scale)
//   3 hl_pos      per positive: decode prediction and target, aligned 3-D IoU -> IoU-prediction smooth-L1 (+ gradient), ODIoU
//                 term with its float64 forward-mode gradient chained through the box decoding (student)
//   4 hl_match    per candidate student box (one wave each): BEV IoU against every teacher candidate, running max / first argmax
//   5 hl_cons     per sample: rows with max IoU > 0.7 -> the three smooth-L1 consistency terms, gradients ADDED at the matched anchors
//   6 hl_final    ordered sums of all partials -> the loss and the log record
// Summation order is fixed (block partials in double, summed in index order): deterministic, no float atomics.
#include "geom.hpp"
#include "odiou_core.hpp"
#include "sessd_hip_types.h"

namespace {

constexpr int NT = 256;
constexpr int NPART = 12;  // per-block partial sums: cls, cls_pos, cls_neg, dir, loc[7], number of negatives
constexpr int REC = 64;    // floats of the log record

struct Work {
  int* blk_cnt;       // [2][B][nblk][2]  positives, consistency candidates of a block
  double* part;       // [2][B][nblk][NPART]
  int* pos_list;      // [2][pos_cap]     flat anchor b * A + a, ascending
  int* counts;        // [0..1] positives of the batch per network | [2 + s*B + b] per sample | [2 + 2B + s*B + b] candidates | [2 + 4B] err
  float* pos_terms;   // [2][pos_cap][2]  weighted IoU-prediction term, weighted ODIoU term
  float* cons_box;    // [2][B][K][7]
  int* cons_anchor;   // [2][B][K]
  float* cons_cls;    // [2][B][K] logits
  float* cons_iou;    // [2][B][K] IoU predictions
  float* row_max;     // [B][K]
  int* row_arg;       // [B][K]
  double* cons_part;  // [B][4]  box, cls, iou term of the sample (already / n1), n1
};

__device__ __forceinline__ int load_label(const void* labels, int i64, size_t i) {
  return i64 ? (int)((const long long*)labels)[i] : ((const int*)labels)[i];
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// second_box_decode (box_torch_ops.py:112-146): code (7) + anchor (7) -> box [x, y, z, w, l, h, r]
__device__ __forceinline__ void decode_box(const float* code, const float* an, float* out) {
  const float diag = sqrtf(an[4] * an[4] + an[3] * an[3]);
  out[0] = code[0] * diag + an[0];
  out[1] = code[1] * diag + an[1];
  out[2] = code[2] * an[5] + an[2];
  out[3] = expf(code[3]) * an[3];
  out[4] = expf(code[4]) * an[4];
  out[5] = expf(code[5]) * an[5];
  out[6] = code[6] + an[6];
}

// d box_k / d code_k of the decoding (diagonal Jacobian)
__device__ __forceinline__ void decode_jac(const float* box, const float* an, float* jac) {
  const float diag = sqrtf(an[4] * an[4] + an[3] * an[3]);
  jac[0] = diag; jac[1] = diag; jac[2] = an[5];
  jac[3] = box[3]; jac[4] = box[4]; jac[5] = box[5];
  jac[6] = 1.0f;
}

// WeightedSmoothL1Loss element (losses.py:188-193), sigma^2 = s2: value and derivative with respect to diff
__device__ __forceinline__ float sl1(float diff, float s2, float* d) {
  const float a = fabsf(diff);
  if (a <= 1.0f / s2) {
    *d = s2 * diff;
    const float t = a * sqrtf(s2);
    return 0.5f * (t * t);
  }
  *d = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
  return a - 0.5f / s2;
}

// the consistency filter of one anchor (mg_head_sessd.py:653-661): score >= thresh and the decoded centre inside the range;
// ONE function for the counting and the compacting launch: their decisions must be the same bits (a disagreement would make
// two candidates share a slot). NOT inlined -- both kernels call the one compiled body, so the expf lowering and the
// contractions cannot differ between them -- and no multiply-add is contracted (round-4 advisor finding).
__device__ __noinline__ bool cons_candidate(float cls_logit, float c0, float c1, float c2, float a0, float a1, float a2, float a3,
                                            float a4, float a5, float thresh, float lo0, float lo1, float lo2, float hi0, float hi1,
                                            float hi2) {
  if (!(sigmoid_f(cls_logit) >= thresh)) return false;
  // every product and sum rounded on its own, as the torch ops of box_torch_ops.second_box_decode round them
  const float diag = sqrtf(__fadd_rn(__fmul_rn(a4, a4), __fmul_rn(a3, a3)));
  const float x = __fadd_rn(__fmul_rn(c0, diag), a0), y = __fadd_rn(__fmul_rn(c1, diag), a1), z = __fadd_rn(__fmul_rn(c2, a5), a2);
  return x >= lo0 && y >= lo1 && z >= lo2 && x <= hi0 && y <= hi1 && z <= hi2;
}
__device__ __forceinline__ bool cons_candidate(float cls_logit, const float* code, const float* an0, const sessd_head_loss_cfg_t& P) {
  return cons_candidate(cls_logit, code[0], code[1], code[2], an0[0], an0[1], an0[2], an0[3], an0[4], an0[5], P.score_thresh,
                        P.center_range[0], P.center_range[1], P.center_range[2], P.center_range[3], P.center_range[4],
                        P.center_range[5]);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the block (NT threads), result valid in every thread; `sm` holds NT / 64 doubles
template <int N>
__device__ __forceinline__ double block_sum_d(double v, double* sm) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < N / 64; ++w) t += sm[w];
  return t;
}

template <int N>
__device__ __forceinline__ int block_sum_i(int v, int* sm) {
  v = sessd_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < N / 64; ++w) t += sm[w];
  return t;
}

// ---------------------------------------------------------------------------------------------------------------- launch 1
__global__ __launch_bounds__(NT) void hl_count_kernel(sessd_head_loss_net_t S, sessd_head_loss_net_t T, sessd_head_loss_cfg_t P,
                                                       const float* __restrict__ anchors0, Work W, int nblk) {
  __shared__ int sm[NT / 64];
  const int s = blockIdx.z, b = blockIdx.y, blk = blockIdx.x;
  const sessd_head_loss_net_t& N = s ? T : S;
  const int a = blk * NT + threadIdx.x;
  int pos = 0, cons = 0;
  if (a < P.num_anchors) {
    const size_t i = (size_t)b * P.num_anchors + a;
    pos = load_label(N.labels, P.labels_i64, i) > 0;
    float code[3], an0[7];
#pragma unroll
    for (int k = 0; k < 3; ++k) code[k] = N.box[i * 7 + k];
#pragma unroll
    for (int k = 0; k < 7; ++k) an0[k] = anchors0[(size_t)a * 7 + k];
    cons = cons_candidate(N.cls[i], code, an0, P);
  }
  const int npos = block_sum_i<NT>(pos, sm);
  const int ncons = block_sum_i<NT>(cons, sm);
  if (threadIdx.x == 0) {
    int* c = W.blk_cnt + (((size_t)s * P.batch + b) * nblk + blk) * 2;
    c[0] = npos;
    c[1] = ncons;
    if (s == 0 && b == 0 && blk == 0) W.counts[2 + 4 * P.batch] = 0;  // the overflow flags of this call
  }
}

// ---------------------------------------------------------------------------------------------------------------- launch 2
__global__ __launch_bounds__(NT) void hl_anchor_kernel(sessd_head_loss_net_t S, sessd_head_loss_net_t T, sessd_head_loss_cfg_t P,
                                                        const float* __restrict__ anchors0, const float* __restrict__ trans, Work W,
                                                        int nblk, float* __restrict__ g_box, float* __restrict__ g_cls,
                                                        float* __restrict__ g_dir, float* __restrict__ g_iou) {
  __shared__ int smi[NT / 64];
  __shared__ double smd[NT / 64];
  const int s = blockIdx.z, b = blockIdx.y, blk = blockIdx.x, B = P.batch, A = P.num_anchors;
  const sessd_head_loss_net_t& N = s ? T : S;
  // ---- positives before this block in the whole batch (flat order), positives of this sample, candidates before this block /
  // of this sample: from the per-block counts of launch 1 (<= B * nblk pairs, ~5 per thread)
  int pos_before = 0, pos_sample = 0, pos_all = 0, cons_before = 0, cons_sample = 0;
  {
    const int2* cnt = (const int2*)(W.blk_cnt + (size_t)s * B * nblk * 2);
    const int mine = b * nblk + blk;
    for (int i = threadIdx.x; i < B * nblk; i += NT) {
      const int2 c = cnt[i];
      const bool same = i >= b * nblk && i < (b + 1) * nblk;
      pos_all += c.x;
      if (i < mine) pos_before += c.x;
      if (same) {
        pos_sample += c.x;
        cons_sample += c.y;
        if (i < mine) cons_before += c.y;
      }
    }
    pos_before = block_sum_i<NT>(pos_before, smi);
    pos_sample = block_sum_i<NT>(pos_sample, smi);
    pos_all = block_sum_i<NT>(pos_all, smi);
    cons_before = block_sum_i<NT>(cons_before, smi);
    cons_sample = block_sum_i<NT>(cons_sample, smi);
  }
  if (threadIdx.x == 0 && blk == 0) {
    if (b == 0) W.counts[s] = pos_all;
    W.counts[2 + s * B + b] = pos_sample;
    W.counts[2 + 2 * B + s * B + b] = cons_sample;
  }
  const float norm = fmaxf((float)pos_sample, 1.0f);  // NormByNumPositives (mg_head_sessd.py:548-551)
  const int a = blk * NT + threadIdx.x;
  const bool live = a < A;
  const size_t i = (size_t)b * A + (live ? a : 0);
  int label = -1;
  float x = 0.f;
  float code[7], an0[7];
  if (live) {
    label = load_label(N.labels, P.labels_i64, i);
    x = N.cls[i];
#pragma unroll
    for (int k = 0; k < 7; ++k) { code[k] = N.box[i * 7 + k]; an0[k] = anchors0[(size_t)a * 7 + k]; }
  }
  const bool pos = label > 0, neg = label == 0;
  double part[NPART];
#pragma unroll
  for (int k = 0; k < NPART; ++k) part[k] = 0.0;
  float gc = 0.f, gd0 = 0.f, gd1 = 0.f;
  if (live) {
    // ---- focal classification loss (losses.py:364-418) with targets = label * cared (mg_head_sessd.py:724)
    const float cls_w = ((neg ? P.neg_cls_weight : 0.f) + (pos ? P.pos_cls_weight : 0.f)) / norm;
    const float t = pos ? (float)label : 0.f;
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    const float p = sigmoid_f(x);
    const float pt = t * p + (1.f - t) * (1.f - p);
    const float base = 1.f - pt;
    const float mod = P.focal_gamma == 2.0f ? base * base : powf(base, P.focal_gamma);
    const float dmod = P.focal_gamma == 2.0f ? 2.f * base : (base > 0.f ? P.focal_gamma * powf(base, P.focal_gamma - 1.f) : 0.f);
    const float aw = t * P.focal_alpha + (1.f - t) * (1.f - P.focal_alpha);
    const float fl = mod * aw * ce * cls_w;
    const float dpt = (2.f * t - 1.f) * p * (1.f - p);
    gc = cls_w * aw * (dmod * (-dpt) * ce + mod * (p - t)) * (P.cls_loss_weight / (float)B);
    part[0] = fl;
    part[1] = pos ? fl : 0.f;
    part[2] = neg ? fl : 0.f;
    part[11] = neg ? 1.0 : 0.0;
  }
  if (live && pos) {
    const float reg_w = 1.0f / norm;
    float tg[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) tg[k] = N.reg_targets[i * 7 + k];
    // ---- direction classifier (mg_head_sessd.py:62-76,746-752; softmax cross entropy losses.py:489-531)
    const float an_r = N.anchors[i * 7 + 6];
    const int dt = ((tg[6] + an_r) - P.direction_offset) > 0.f ? 1 : 0;
    const float l0 = N.dir[i * 2], l1 = N.dir[i * 2 + 1];
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float lse = m + logf(e0 + e1);
    part[3] = (lse - (dt ? l1 : l0)) * reg_w;
    const float sc = reg_w * P.dir_loss_weight / (float)B;
    gd0 = (e0 / (e0 + e1) - (dt ? 0.f : 1.f)) * sc;
    gd1 = (e1 / (e0 + e1) - (dt ? 1.f : 0.f)) * sc;
    // ---- smooth-L1 localisation on sin-encoded yaw: logged only (mg_head_sessd.py:39-44,735-739,756)
    const float s2 = P.smooth_l1_sigma * P.smooth_l1_sigma;
    float d_;
#pragma unroll
    for (int k = 0; k < 6; ++k) part[4 + k] = sl1(code[k] - tg[k], s2, &d_) * reg_w;
    part[10] = sl1(sinf(code[6]) * cosf(tg[6]) - cosf(code[6]) * sinf(tg[6]), s2, &d_) * reg_w;
  }
  // ---- ordered compaction: positives of the batch, consistency candidates of the sample
  const bool cand = live && cons_candidate(x, code, an0, P);
  int tot;
  const int prank = sessd_block_exscan<NT>(pos ? 1 : 0, smi, &tot);
  const int crank = sessd_block_exscan<NT>(cand ? 1 : 0, smi, &tot);
  if (pos) {
    const int at = pos_before + prank;
    if (at < P.pos_capacity) W.pos_list[(size_t)s * P.pos_capacity + at] = (int)i;
    else atomicOr(&W.counts[2 + 4 * B], 1);
  }
  if (cand) {
    const int at = cons_before + crank;
    if (at < P.cons_capacity) {
      const size_t o = ((size_t)s * B + b) * P.cons_capacity + at;
      float bx[7];
      decode_box(code, an0, bx);
      if (s == 1) {  // teacher box into the student's frame (mg_head_sessd.py:670-674): flip, rotation about z, scale
        const float* tr = trans + (size_t)b * 5;  // [flipped, cos, sin, rotation, scale]
        if (tr[0] != 0.f) {
          bx[1] = -bx[1];
          bx[6] = -bx[6] + 3.14159265358979323846f;
        }
        const float rx = bx[0] * tr[1] + bx[1] * tr[2], ry = -bx[0] * tr[2] + bx[1] * tr[1];
        bx[0] = rx; bx[1] = ry;
        bx[6] += tr[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) bx[k] *= tr[4];
      }
#pragma unroll
      for (int k = 0; k < 7; ++k) W.cons_box[o * 7 + k] = bx[k];
      W.cons_anchor[o] = a;
      W.cons_cls[o] = x;
      W.cons_iou[o] = N.iou[i];
    } else {
      atomicOr(&W.counts[2 + 4 * B], 2);
    }
  }
  // ---- gradients of the student's per-anchor terms; the box / IoU-prediction gradients of this anchor start at zero
  // (launches 3 and 5 write / add the positives' and the matched boxes')
  if (live && s == 0) {
    g_cls[i] = gc;
    g_dir[i * 2] = gd0;
    g_dir[i * 2 + 1] = gd1;
    g_iou[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) g_box[i * 7 + k] = 0.f;
  }
  double* out = W.part + (((size_t)s * B + b) * nblk + blk) * NPART;
#pragma unroll
  for (int k = 0; k < NPART; ++k) {
    const double v = block_sum_d<NT>(part[k], smd);
    if (threadIdx.x == 0) out[k] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------- launch 3
struct PosLds {
  float px[SESSD_IOU_MAXPTS][64];
  float py[SESSD_IOU_MAXPTS][64];
  float pa[SESSD_IOU_MAXPTS][64];
};

// eight lanes per positive: lane c < 7 carries d ODIoU / d q[c] (odiou_core.hpp), lane 0 also does the IoU-prediction term
__global__ __launch_bounds__(64) void hl_pos_kernel(sessd_head_loss_net_t S, sessd_head_loss_net_t T, sessd_head_loss_cfg_t P, Work W,
                                                     float* __restrict__ g_box, float* __restrict__ g_iou) {
  __shared__ PosLds L;
  const int s = blockIdx.y, B = P.batch, A = P.num_anchors;
  const sessd_head_loss_net_t& N = s ? T : S;
  const int n = min(W.counts[s], P.pos_capacity);
  const int r = blockIdx.x * 8 + (threadIdx.x >> 3), c = threadIdx.x & 7;
  if (r >= n) return;
  const size_t i = (size_t)W.pos_list[(size_t)s * P.pos_capacity + r];
  const int b = (int)(i / A);
  const float reg_w = 1.0f / fmaxf((float)W.counts[2 + s * B + b], 1.0f);
  float code[7], tg[7], an[7], q[7], g[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { code[k] = N.box[i * 7 + k]; tg[k] = N.reg_targets[i * 7 + k]; an[k] = N.anchors[i * 7 + k]; }
  decode_box(code, an, q);
  decode_box(tg, an, g);
  float* terms = W.pos_terms + ((size_t)s * P.pos_capacity + r) * 2;
  if (c == 0) {
    // ---- IoU-prediction target 2 * IoU3D(q, g) - 1 (iou3d_utils.py:197-252: rotated BEV overlap x height overlap)
    sessd_rect RQ, RG;
    sessd_rect_init(RQ, q[0] - q[3] / 2, q[1] - q[4] / 2, q[0] + q[3] / 2, q[1] + q[4] / 2, q[6]);
    sessd_rect_init(RG, g[0] - g[3] / 2, g[1] - g[4] / 2, g[0] + g[3] / 2, g[1] + g[4] / 2, g[6]);
    sessd_ptlist PL = {&L.px[0][threadIdx.x], &L.py[0][threadIdx.x], &L.pa[0][threadIdx.x], 64};
    const float ov = sessd_rect_overlap_f32(RQ, RG, PL);
    const float qlo = q[2] - q[5] / 2, qhi = q[2] + q[5] / 2, glo = g[2] - g[5] / 2, ghi = g[2] + g[5] / 2;
    const float oh = fmaxf(fminf(qhi, ghi) - fmaxf(qlo, glo), 0.f);
    const float o3 = ov * oh;
    const float iou3d = o3 / fmaxf(q[3] * q[4] * q[5] + g[3] * g[4] * g[5] - o3, 1e-7f);
    const float target = 2.f * iou3d - 1.f;
    const float s2 = 9.0f;  // loss_iou_pred is built with sigma = 3 whatever the config says (mg_head_sessd.py:431,767)
    float d;
    const float v = sl1(N.iou[i] - target, s2, &d);
    terms[0] = v * reg_w;
    if (s != 0) terms[1] = 0.f;
    else g_iou[i] = d * reg_w / (float)B;
  }
  if (s != 0) return;
  // ---- ODIoU (odious.py:837-900): loss 2 * sum(w * term) / B; gradient through the box decoding (diagonal Jacobian)
  double gd[7], qd[7], term, grad;
#pragma unroll
  for (int k = 0; k < 7; ++k) { gd[k] = (double)g[k]; qd[k] = (double)q[k]; }
  const int comp = c < 7 ? c : 6;
  odiou_eval(gd, qd, comp, &term, &grad);
  if (c == 0) terms[1] = (float)term * reg_w;
  if (c < 7) {
    float jac[7];
    decode_jac(q, an, jac);
    const float jc = c == 0 ? jac[0] : c == 1 ? jac[1] : c == 2 ? jac[2] : c == 3 ? jac[3] : c == 4 ? jac[4] : c == 5 ? jac[5] : jac[6];
    const float sc = 2.0f * reg_w / (float)B;
    g_box[i * 7 + c] = (float)grad * sc * jc;
  }
}

// ---------------------------------------------------------------------------------------------------------------- launch 4
struct MatchLds {
  float px[SESSD_IOU_MAXPTS][NT];
  float py[SESSD_IOU_MAXPTS][NT];
  float pa[SESSD_IOU_MAXPTS][NT];
};

// one wave per candidate student box of sample blockIdx.y: max and FIRST argmax of the BEV IoU over the teacher's candidates
__global__ __launch_bounds__(NT) void hl_match_kernel(sessd_head_loss_cfg_t P, Work W) {
  __shared__ MatchLds L;
  const int b = blockIdx.y, B = P.batch, K = P.cons_capacity;
  const int ns = min(W.counts[2 + 2 * B + b], K), nt = min(W.counts[2 + 2 * B + B + b], K);
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= ns || nt == 0) return;
  const float* sb = W.cons_box + ((size_t)b * K + row) * 7;
  sessd_rect RS;
  sessd_rect_init(RS, sb[0] - sb[3] / 2, sb[1] - sb[4] / 2, sb[0] + sb[3] / 2, sb[1] + sb[4] / 2, sb[6]);
  sessd_ptlist PL = {&L.px[0][threadIdx.x], &L.py[0][threadIdx.x], &L.pa[0][threadIdx.x], NT};
  float best = -1.0f;
  int arg = 0x7fffffff;
  for (int j = lane; j < nt; j += 64) {
    const float* tb = W.cons_box + (((size_t)B + b) * K + j) * 7;
    sessd_rect RT;
    sessd_rect_init(RT, tb[0] - tb[3] / 2, tb[1] - tb[4] / 2, tb[0] + tb[3] / 2, tb[1] + tb[4] / 2, tb[6]);
    const float so = sessd_rect_overlap_f32(RS, RT, PL);
    const float sa = (RS.x2 - RS.x1) * (RS.y2 - RS.y1), sbb = (RT.x2 - RT.x1) * (RT.y2 - RT.y1);
    const float iou = so / fmaxf(sa + sbb - so, SESSD_IOU_EPS);
    if (iou > best) { best = iou; arg = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  if (lane == 0) {
    W.row_max[(size_t)b * K + row] = best;
    W.row_arg[(size_t)b * K + row] = arg;
  }
}

// ---------------------------------------------------------------------------------------------------------------- launch 5
constexpr int NTC = 1024;

__global__ __launch_bounds__(NTC) void hl_cons_kernel(sessd_head_loss_cfg_t P, const float* __restrict__ anchors0,
                                                       const float* __restrict__ cons_weight, Work W, float* __restrict__ g_box,
                                                       float* __restrict__ g_cls, float* __restrict__ g_iou) {
  __shared__ int smi[NTC / 64];
  __shared__ double smd[NTC / 64];
  const int b = blockIdx.x, B = P.batch, K = P.cons_capacity, A = P.num_anchors;
  const int ns = min(W.counts[2 + 2 * B + b], K), nt = min(W.counts[2 + 2 * B + B + b], K);
  double* out = W.cons_part + (size_t)b * 4;
  int n1 = 0;
  if (nt > 0)
    for (int r = threadIdx.x; r < ns; r += NTC) n1 += W.row_max[(size_t)b * K + r] > P.match_iou_thresh ? 1 : 0;
  n1 = block_sum_i<NTC>(n1, smi);
  if (n1 == 0) {  // mg_head_sessd.py:667,679-680: nothing matched in this sample
    if (threadIdx.x < 4) out[threadIdx.x] = 0.0;
    return;
  }
  const float cw = cons_weight[0];
  const float scale = cw / ((float)n1 * (float)B);
  // the score / IoU consistency losses are built with sigma = 3 whatever the config says (mg_head_sessd.py:488-491); the box
  // term goes through self.loss_reg, i.e. the config's loss_bbox.sigma (nn_distance :594-597)
  const float s2 = 9.0f, s2_box = P.smooth_l1_sigma * P.smooth_l1_sigma;
  double l_box = 0.0, l_cls = 0.0, l_iou = 0.0;
  for (int r = threadIdx.x; r < ns; r += NTC) {
    if (!(W.row_max[(size_t)b * K + r] > P.match_iou_thresh)) continue;
    const int j = W.row_arg[(size_t)b * K + r];
    const size_t os = (size_t)b * K + r, ot = ((size_t)B + b) * K + j;
    const float* sb = W.cons_box + os * 7;
    const float* tb = W.cons_box + ot * 7;
    const int a = W.cons_anchor[os];
    const size_t i = (size_t)b * A + a;
    float an[7], jac[7], gb[7], d;
#pragma unroll
    for (int k = 0; k < 7; ++k) an[k] = anchors0[(size_t)a * 7 + k];
    decode_jac(sb, an, jac);
    // box term: smooth-L1 over [x, y, z, w, l, h, sin(a - b) split as sin a cos b / cos a sin b], summed / 7 (:594-595)
    float lb = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      lb += sl1(sb[k] - tb[k], s2_box, &d);
      gb[k] = d / 7.0f;
    }
    const float sa = sinf(sb[6]), ca = cosf(sb[6]), st = sinf(tb[6]), ct = cosf(tb[6]);
    lb += sl1(sa * ct - ca * st, s2_box, &d);
    gb[6] = d * (ca * ct + sa * st) / 7.0f;
    l_box += (double)(lb / 7.0f);
#pragma unroll
    for (int k = 0; k < 7; ++k) g_box[i * 7 + k] += gb[k] * jac[k] * scale;
    // score term (:685-686)
    const float ps = sigmoid_f(W.cons_cls[os]), pt = sigmoid_f(W.cons_cls[ot]);
    l_cls += (double)sl1(ps - pt, s2, &d);
    g_cls[i] += d * ps * (1.f - ps) * scale;
    // IoU-prediction term (:690-692)
    const float is = (W.cons_iou[os] + 1.f) * 0.5f, it = (W.cons_iou[ot] + 1.f) * 0.5f;
    l_iou += (double)sl1(is - it, s2, &d);
    g_iou[i] += d * 0.5f * scale;
  }
  l_box = block_sum_d<NTC>(l_box, smd);
  l_cls = block_sum_d<NTC>(l_cls, smd);
  l_iou = block_sum_d<NTC>(l_iou, smd);
  if (threadIdx.x == 0) {
    out[0] = l_box / n1;
    out[1] = l_cls / n1;
    out[2] = l_iou / n1;
    out[3] = (double)n1;
  }
}

// ---------------------------------------------------------------------------------------------------------------- launch 6
__global__ __launch_bounds__(NT) void hl_final_kernel(sessd_head_loss_cfg_t P, const float* __restrict__ cons_weight, Work W, int nblk,
                                                       float* __restrict__ record) {
  __shared__ double smd[NT / 64];
  __shared__ double tot[2][NPART + 2];
  const int B = P.batch;
  for (int s = 0; s < 2; ++s) {
    double acc[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) acc[k] = 0.0;
    // per-thread strided, then a fixed tree: the order depends on (B, nblk) only
    for (int i = threadIdx.x; i < B * nblk; i += NT) {
      const double* p = W.part + ((size_t)s * B * nblk + i) * NPART;
#pragma unroll
      for (int k = 0; k < NPART - 1; ++k) acc[k] += p[k];
      if (i < nblk) acc[NPART - 1] += p[NPART - 1];  // negatives of sample 0 only (num_neg, mg_head_sessd.py:794)
    }
    const int n = min(W.counts[s], P.pos_capacity);
    double ip = 0.0, od = 0.0;
    for (int r = threadIdx.x; r < n; r += NT) {
      const float* t = W.pos_terms + ((size_t)s * P.pos_capacity + r) * 2;
      ip += (double)t[0];
      od += (double)t[1];
    }
#pragma unroll
    for (int k = 0; k < NPART; ++k) {
      const double v = block_sum_d<NT>(acc[k], smd);
      if (threadIdx.x == 0) tot[s][k] = v;
    }
    ip = block_sum_d<NT>(ip, smd);
    od = block_sum_d<NT>(od, smd);
    if (threadIdx.x == 0) { tot[s][NPART] = ip; tot[s][NPART + 1] = od; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double invB = 1.0 / (double)B;
  double cbox = 0.0, ccls = 0.0, ciou = 0.0, matched = 0.0;
  for (int b = 0; b < B; ++b) {
    const double* c = W.cons_part + (size_t)b * 4;
    cbox += c[0]; ccls += c[1]; ciou += c[2]; matched += c[3];
  }
  const double cons = (cbox + ccls + ciou) * invB;  // :702
  for (int k = 0; k < REC; ++k) record[k] = 0.f;
  for (int s = 0; s < 2; ++s) {
    const double* t = tot[s];
    const double cls_red = P.cls_loss_weight * t[0] * invB, dir = P.dir_loss_weight * t[3] * invB;
    double loc = 0.0;
    for (int k = 0; k < 7; ++k) loc += t[4 + k];
    const double loc_red = P.loc_loss_weight * loc * invB, iou_pred = t[NPART] * invB, odiou = 2.0 * t[NPART + 1] * invB;
    float* r = record + (s ? 24 : 0);
    if (s == 0) {
      const double sup = cls_red + odiou + dir + iou_pred;                 // mg_head_sessd.py:780
      r[0] = (float)(sup + cons * (double)cons_weight[0]);                 // trainer_sessd.py:267
      r[1] = (float)sup;
      r[6] = (float)odiou;
      r[7] = (float)cons;
      r[19] = (float)(cbox * invB); r[20] = (float)(ccls * invB); r[21] = (float)(ciou * invB); r[22] = (float)matched;
    } else {
      r[0] = (float)(cls_red + dir + iou_pred);                            // :867 loss_ema
    }
    r[2] = (float)cls_red;
    r[3] = (float)loc_red;
    r[4] = (float)dir;
    r[5] = (float)iou_pred;
    r[8] = (float)(t[1] * invB / P.pos_cls_weight);
    r[9] = (float)(t[2] * invB / P.neg_cls_weight);
    for (int k = 0; k < 7; ++k) r[10 + k] = (float)(t[4 + k] * invB);
    r[17] = (float)W.counts[2 + s * B];   // positives of sample 0
    r[18] = (float)t[NPART - 1];          // negatives of sample 0
  }
  record[48] = (float)W.counts[2 + 4 * B];  // bit 0: positives beyond pos_capacity dropped, bit 1: candidates beyond cons_capacity
  record[49] = (float)W.counts[0];
  record[50] = (float)W.counts[1];
  int cs = 0, ct = 0;
  for (int b = 0; b < B; ++b) { cs = max(cs, W.counts[2 + 2 * B + b]); ct = max(ct, W.counts[2 + 2 * B + B + b]); }
  record[51] = (float)cs;  // most consistency candidates in one sample: student, teacher
  record[52] = (float)ct;
}

size_t carve(char*& p, size_t bytes) {
  const size_t o = (size_t)p;
  p += sessd_align(bytes, 256);
  return o;
}

bool cfg_ok(const sessd_head_loss_cfg_t* c) {
  return c && c->batch >= 1 && c->batch <= 64 && c->num_anchors >= 1 && c->pos_capacity >= 1 && c->cons_capacity >= 1 &&
         (long long)c->batch * c->num_anchors < (1ll << 31) && c->smooth_l1_sigma > 0.f && c->pos_cls_weight > 0.f &&
         c->neg_cls_weight > 0.f;
}

Work layout(const sessd_head_loss_cfg_t* c, void* base, size_t* total) {
  const int B = c->batch, nblk = sessd_divup(c->num_anchors, NT), K = c->cons_capacity, PC = c->pos_capacity;
  char* p = (char*)base;
  Work W;
  W.blk_cnt = (int*)carve(p, (size_t)2 * B * nblk * 2 * 4);
  W.part = (double*)carve(p, (size_t)2 * B * nblk * NPART * 8);
  W.pos_list = (int*)carve(p, (size_t)2 * PC * 4);
  W.counts = (int*)carve(p, (size_t)(2 + 4 * B + 1) * 4);
  W.pos_terms = (float*)carve(p, (size_t)2 * PC * 2 * 4);
  W.cons_box = (float*)carve(p, (size_t)2 * B * K * 7 * 4);
  W.cons_anchor = (int*)carve(p, (size_t)2 * B * K * 4);
  W.cons_cls = (float*)carve(p, (size_t)2 * B * K * 4);
  W.cons_iou = (float*)carve(p, (size_t)2 * B * K * 4);
  W.row_max = (float*)carve(p, (size_t)B * K * 4);
  W.row_arg = (int*)carve(p, (size_t)B * K * 4);
  W.cons_part = (double*)carve(p, (size_t)B * 4 * 8);
  *total = (size_t)(p - (char*)base);
  return W;
}

}  // namespace

extern "C" {

size_t sessd_head_loss_workspace_bytes(const sessd_head_loss_cfg_t* cfg) {
  if (!cfg_ok(cfg)) return 0;
  size_t total;
  layout(cfg, nullptr, &total);
  return total;
}

int sessd_head_loss(const sessd_head_loss_cfg_t* cfg, const sessd_head_loss_net_t* student, const sessd_head_loss_net_t* teacher,
                    const float* anchors0, const float* transformation, const float* consistency_weight, float* grad_box,
                    float* grad_cls, float* grad_dir, float* grad_iou, float* record, void* workspace, size_t workspace_bytes,
                    hipStream_t stream) {
  if (!cfg_ok(cfg) || !student || !teacher || !anchors0 || !transformation || !consistency_weight || !grad_box || !grad_cls ||
      !grad_dir || !grad_iou || !record || !workspace)
    return SESSD_EINVAL;
  if (((size_t)workspace & 255) != 0) return SESSD_EINVAL;
  size_t need;
  const Work W = layout(cfg, workspace, &need);
  if (workspace_bytes < need) return SESSD_EWORKSPACE;
  const sessd_head_loss_cfg_t P = *cfg;
  const int B = P.batch, nblk = sessd_divup(P.num_anchors, NT);
  SESSD_LAUNCH(hl_count_kernel, dim3(nblk, B, 2), dim3(NT), 0, stream, *student, *teacher, P, anchors0, W, nblk);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(hl_anchor_kernel, dim3(nblk, B, 2), dim3(NT), 0, stream, *student, *teacher, P, anchors0, transformation, W, nblk,
               grad_box, grad_cls, grad_dir, grad_iou);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(hl_pos_kernel, dim3(sessd_divup(P.pos_capacity, 8), 2), dim3(64), 0, stream, *student, *teacher, P, W, grad_box,
               grad_iou);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(hl_match_kernel, dim3(sessd_divup(P.cons_capacity, NT / 64), B), dim3(NT), 0, stream, P, W);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(hl_cons_kernel, dim3(B), dim3(NTC), 0, stream, P, anchors0, consistency_weight, W, grad_box, grad_cls, grad_iou);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(hl_final_kernel, dim3(1), dim3(NT), 0, stream, P, consistency_weight, W, nblk, record);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
