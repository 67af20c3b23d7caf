// DI-NMS: the IoU-weighted rotated NMS of det3d/ops/nms/nms_cpu.h:173-384 (IOU_weighted_rotate_non_max_suppression_cpu, the
// pybind core behind det3d/core/bbox/box_torch_ops.py:552-621 rotate_weighted_nms; SURVEY 8f row 4 lists it as the alternative
// post-processor) on the device.
//
// Two launches. (1) overlap[i][j] = |A_i n A_j| / |A_i u A_j| of the rotated footprints for all pairs (float64 clipper of the
// rotated NMS, rounded to float like the reference's float instantiation); the reference recomputes these N polygon
// intersections with boost::geometry inside every pass of its loop. (2) ONE workgroup of 1024 threads, thread j = box j
// (N <= 1024 = the pre_max_size the post-processor keeps), runs the reference's sequential loop:
//   pick the unsuppressed box A with the largest original score (lowest index on ties), mark it;
//   every box j with overlap(A, j) > 0 and A's label adds overlap * iou_pred[j] to cnt; those above suppressed_thresh also feed
//   the weighted box average (weight exp(-(1 - overlap)^2 / sigma^2(|A|)) * iou_pred[j]) and the maximum normalised score;
//   unsuppressed boxes with stand-up IoU > 0 and overlap >= suppressed_thresh are suppressed;
//   cnt > cnt_thresh: A is kept with the averaged box; otherwise this pass's suppressions are undone.
// The nine sums of a pass (cnt, 7 weighted coordinates, the weight) are reduced in a FIXED tree (wave shuffles, then the 16 wave
// results in order): deterministic, but not the reference's left-to-right float sum -- tests compare with tolerance.
#include "common.hpp"
#include "geom.hpp"

namespace {

constexpr int DN = 1024;

__global__ __launch_bounds__(256) void di_overlap_kernel(const float* __restrict__ corners, int n, float* __restrict__ overlap) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (size_t)n * n) return;
  const int i = (int)(id / n), j = (int)(id - (size_t)i * n);
  const float* pi = corners + (size_t)i * 8;
  const float* pj = corners + (size_t)j * 8;
  float v = 0.f;
  const double inter = sessd_quad_inter_area_green(pi, pj);
  if (inter > 0) {
    double px[4], py[4], qx[4], qy[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { px[q] = pi[2 * q]; py[q] = pi[2 * q + 1]; qx[q] = pj[2 * q]; qy[q] = pj[2 * q + 1]; }
    const float ia = (float)inter;
    const float ua = (float)(fabs(sessd_poly_area2(px, py, 4)) * 0.5 + fabs(sessd_poly_area2(qx, qy, 4)) * 0.5 - inter);
    v = ua > 0.f ? ia / ua : 0.f;
  }
  overlap[id] = v;
}

struct DiArgs {
  const float* boxes;        // (n, 7)
  const float* overlap;      // (n, n)
  const float* standup;      // (n, n)
  const float* scores;       // (n)
  const float* iou_preds;    // (n)
  const int* labels;
  const int* dirs;
  const float* anchors;      // (n, anchor_stride) or null
  int anchor_stride, n, n_interval, centerness_c;
  float cnt_thresh, suppressed_thresh;
  float interval[8], sigma_sq[8];
  float* boxes_ret;          // (n, 7)
  float* scores_ret;
  int* labels_ret;
  int* dirs_ret;
  int* keep;
  int* n_keep;
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(DN) void di_select_kernel(DiArgs A) {
  __shared__ float s_red[16][10];
  __shared__ unsigned long long s_key[16];
  __shared__ float s_bc[12];
  __shared__ float s_terms[1024];  // the pass's non-zero cnt terms in box order
  __shared__ int s_nz[16];
  const int j = threadIdx.x, lane = j & 63, wv = j >> 6;
  const int n = A.n;
  const bool live = j < n;
  // ---- normalised scores (nms_cpu.h:233-262)
  float srw = live ? A.scores[j] : 0.f;
  if (A.centerness_c == 1) {
    float c = 0.f;
    if (live) {
      const double dx = (double)(A.boxes[j * 7] - A.anchors[(size_t)j * A.anchor_stride]);
      const double dy = (double)(A.boxes[j * 7 + 1] - A.anchors[(size_t)j * A.anchor_stride + 1]);
      c = (float)exp((double)(float)sqrt(dx * dx + dy * dy));
    }
    float t = wave_sum_f(c);
    if (lane == 0) s_red[wv][0] = t;
    __syncthreads();
    float sum = 0.f;
    for (int w = 0; w < 16; ++w) sum += s_red[w][0];
    __syncthreads();
    if (live) srw *= (1 - c / sum);
  }
  {
    float m = live ? srw : -10000.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) s_red[wv][0] = m;
    __syncthreads();
    float mx = -10000.f;
    for (int w = 0; w < 16; ++w) mx = fmaxf(mx, s_red[w][0]);
    __syncthreads();
    if (j == 0) s_bc[11] = mx;
    srw /= mx;
  }
  const float my_score = live ? A.scores[j] : 0.f;
  const int my_label = live ? A.labels[j] : -1;
  const float my_ioup = live ? A.iou_preds[j] : 0.f;
  float my_box[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) my_box[k] = live ? A.boxes[j * 7 + k] : 0.f;
  bool suppressed = !live;
  int nkeep = 0;
  __syncthreads();
  const float score_max4norm = s_bc[11];
  for (;;) {
    // ---- the unsuppressed box with the largest original score, lowest index on ties (strict > in the reference's scan); a score
    // must exceed -1 to be picked (score_max starts at -1)
    unsigned long long key = 0ull;
    if (!suppressed && my_score > -1.f) {
      unsigned u = __float_as_uint(my_score);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // order-preserving map of the float
      key = ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)j);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o, 64);
      key = other > key ? other : key;
    }
    if (lane == 0) s_key[wv] = key;
    __syncthreads();
    unsigned long long best = 0ull;
    for (int w = 0; w < 16; ++w) best = s_key[w] > best ? s_key[w] : best;
    if (best == 0ull) break;                                    // everything suppressed (uniform)
    const int idx = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    if (j == idx) suppressed = true;
    const float bx = A.boxes[idx * 7], by = A.boxes[idx * 7 + 1];
    const float dist2origin = (float)sqrt((double)bx * bx + (double)by * by);
    const int lab = A.labels[idx];
    // ---- this box's contribution
    float cnt = 0.f, wsum = 0.f, avg[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sbox = -1.f;
    bool newly = false;
    if (live) {
      const float ov = A.overlap[(size_t)idx * n + j];
      if (ov > 0.f) {
        const bool same = my_label == lab;
        if (same) cnt = ov * my_ioup;
        if (ov > A.suppressed_thresh && same) {
          sbox = srw;
          float w = 0.f;
          for (int k = 0; k + 1 < A.n_interval; ++k)
            if (dist2origin >= A.interval[k] && dist2origin < A.interval[k + 1]) {
              const double d = 1.0 - (double)ov;
              w = (float)exp(-(d * d) / (double)A.sigma_sq[k]);
            }
          wsum = w * my_ioup;
#pragma unroll
          for (int k = 0; k < 7; ++k) avg[k] = w * my_ioup * my_box[k];
        }
        if (!suppressed && A.standup[(size_t)idx * n + j] > 0.f && ov >= A.suppressed_thresh) {
          suppressed = true;
          newly = true;
        }
      }
    }
    // ---- cnt decides whether A is kept (cnt > cnt_thresh), and everything selected later follows from that: it is summed in the
    // REFERENCE'S ORDER, box 0, 1, 2, ... in float32 (nms_cpu.h: `cnt += overlap * iou_pred[j]` inside the loop over j). Adding
    // zero changes nothing, so only the non-zero terms (a handful: boxes that overlap A) are compacted in box order and one
    // thread adds them up. (Round 2 reduced cnt in a tree: a cnt within an ulp of the threshold could flip a keep / recover.)
    const float cnt_term = cnt;
    const unsigned long long nzb = __ballot(cnt_term != 0.f);
    // ---- fixed-tree reduction of the weight, the 7 coordinates (sum) and the score (max)
    cnt = wave_sum_f(cnt);
    wsum = wave_sum_f(wsum);
#pragma unroll
    for (int k = 0; k < 7; ++k) avg[k] = wave_sum_f(avg[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sbox = fmaxf(sbox, __shfl_xor(sbox, o, 64));
    __syncthreads();   // s_key / s_red of the previous pass are no longer read
    if (lane == 0) {
      s_red[wv][0] = cnt; s_red[wv][1] = wsum; s_red[wv][9] = sbox;
#pragma unroll
      for (int k = 0; k < 7; ++k) s_red[wv][2 + k] = avg[k];
      s_nz[wv] = __popcll(nzb);
    }
    __syncthreads();
    if (cnt_term != 0.f) {
      int off = 0;
      for (int w = 0; w < wv; ++w) off += s_nz[w];
      s_terms[off + __popcll(nzb & ((1ull << lane) - 1ull))] = cnt_term;
    }
    if (j >= 1 && j < 10) {
      float v = s_red[0][j];
      for (int w = 1; w < 16; ++w) v = (j == 9) ? fmaxf(v, s_red[w][j]) : v + s_red[w][j];
      s_bc[j] = v;
    }
    __syncthreads();
    if (j == 0) {
      int total = 0;
      for (int w = 0; w < 16; ++w) total += s_nz[w];
      float c = 0.f;
      for (int t = 0; t < total; ++t) c += s_terms[t];
      s_bc[0] = c;
    }
    __syncthreads();
    const bool kept = s_bc[0] > A.cnt_thresh;
    if (kept) {
      if (j == 0) {
        A.keep[nkeep] = idx;
        A.scores_ret[nkeep] = s_bc[9] * score_max4norm;
        A.labels_ret[nkeep] = lab;
        A.dirs_ret[nkeep] = A.dirs[idx];
      }
      if (j < 7) A.boxes_ret[nkeep * 7 + j] = s_bc[2 + j] / s_bc[1];
      ++nkeep;
    } else if (newly) {
      suppressed = false;   // the boxes this pass suppressed come back; A itself stays marked
    }
    __syncthreads();
  }
  if (j == 0) *A.n_keep = nkeep;
}

}  // namespace

extern "C" {

// Workspace = the (n, n) float overlap matrix.
size_t sessd_di_nms_workspace_bytes(int n) { return n > 0 ? sessd_align((size_t)n * n * 4, 256) : 256; }

// boxes (n,7), corners (n,4,2), standup_iou (n,n) [iou_jit of the stand-up boxes, eps 0], scores / iou_preds (n), labels / dirs (n)
// int32, anchors (n, anchor_stride) or NULL (centerness_c = 0): all device pointers, n <= 1024. sigma_dist_interval (n_interval <= 8
// floats) and sigma_square (n_interval - 1 used) are HOST arrays. Outputs (device, capacity n): averaged boxes, scores, labels,
// directions, kept input indices, and *n_keep.
int sessd_di_nms(const float* boxes, const float* corners, const float* standup_iou, int n, const float* scores,
                 const float* iou_preds, const int* labels, const int* dirs, const float* anchors, int anchor_stride,
                 float cnt_thresh, const float* sigma_dist_interval, int n_interval, const float* sigma_square,
                 float suppressed_thresh, int centerness_c, float* boxes_ret, float* scores_ret, int* labels_ret, int* dirs_ret,
                 int* keep, int* n_keep, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n < 0 || n > DN || n_interval < 0 || n_interval > 8 || (centerness_c == 1 && (!anchors || anchor_stride < 2))) return SESSD_EINVAL;
  if (sessd_di_nms_workspace_bytes(n) > workspace_bytes) return SESSD_EWORKSPACE;
  if (n == 0) {
    SESSD_FILL(n_keep, 0, 1, stream);
    return SESSD_OK;
  }
  float* overlap = (float*)workspace;
  const size_t total = (size_t)n * n;
  SESSD_LAUNCH(di_overlap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, corners, n, overlap);
  SESSD_CHECK_LAUNCH();
  DiArgs A;
  A.boxes = boxes; A.overlap = overlap; A.standup = standup_iou; A.scores = scores; A.iou_preds = iou_preds; A.labels = labels;
  A.dirs = dirs; A.anchors = anchors; A.anchor_stride = anchor_stride; A.n = n; A.n_interval = n_interval;
  A.centerness_c = centerness_c; A.cnt_thresh = cnt_thresh; A.suppressed_thresh = suppressed_thresh;
  for (int k = 0; k < 8; ++k) {
    A.interval[k] = k < n_interval ? sigma_dist_interval[k] : 0.f;
    A.sigma_sq[k] = k + 1 < n_interval ? sigma_square[k] : 1.f;
  }
  A.boxes_ret = boxes_ret; A.scores_ret = scores_ret; A.labels_ret = labels_ret; A.dirs_ret = dirs_ret; A.keep = keep;
  A.n_keep = n_keep;
  SESSD_LAUNCH(di_select_kernel, dim3(1), dim3(DN), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
