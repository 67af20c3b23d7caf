// KITTI average-precision accumulation on the device: the per-frame greedy matching of detections to ground
// truths and the precision / recall bookkeeping of det3d/datasets/kitti/eval.py:174-319 (eval_class_v3), :121-171
// (fused_compute_statistics), :18-37 (get_thresholds) and det3d/datasets/utils/eval.py:144-278 (compute_statistics_jit), which the
// reference runs as numba loops on the host.
// The matching of one frame is a sequential greedy assignment (ground truths in order, each taking the best free detection),
// but frames -- and, in the second pass, the 41 score thresholds -- are independent: one thread per (frame, threshold), the
// frame's few dozen boxes in registers / L1, every comparison in double precision in the reference's order, so the integer
// statistics are identical and the orientation similarity is summed in a fixed order (per frame in matching order, then frames
// in order by one thread per threshold: no float atomics).
//   statistics_kernel  pass 1 (compute_fp = 0): the score of every true positive, per ground-truth slot (NaN = none)
//                      pass 2 (compute_fp = 1): tp / fp / fn / similarity per (frame, threshold)
//   thresholds_kernel  the scores at which the recall crosses the 41 sample levels (one thread: a 40-step scan of a sorted list)
//   reduce_kernel      per threshold the ordered sum over the frames
#include "common.hpp"

namespace {

constexpr int MAXDET = 256;  // detections per frame (bit mask of assigned ones in registers)

struct EvalArgs {
  const double* overlaps;     // per frame a (n_det, n_gt) row-major block at ov_off[f]
  const long long* ov_off;    // (F)
  const int* gt_off;          // (F+1) rows into gt_data / ignored_gt
  const int* dt_off;          // (F+1) rows into dt_data / ignored_det
  const int* dc_off;          // (F+1) rows into dc_boxes
  const double* gt_data;      // (sum n_gt, 5) bbox 4 + alpha
  const double* dt_data;      // (sum n_det, 6) bbox 4 + alpha + score
  const int* ignored_gt;      // (sum n_gt)   0 counts | 1 neutral | -1 other class
  const int* ignored_det;     // (sum n_det)
  const double* dc_boxes;     // (sum n_dc, 4)
  const double* thresholds;   // (T) (pass 2)
  int num_frames, num_thresholds, metric, compute_fp, compute_aos;
  double min_overlap;
};

__device__ __forceinline__ bool bit(const unsigned* m, int j) { return (m[j >> 5] >> (j & 31)) & 1u; }
__device__ __forceinline__ void set(unsigned* m, int j) { m[j >> 5] |= 1u << (j & 31); }

__global__ __launch_bounds__(128) void statistics_kernel(EvalArgs A, double* __restrict__ tp_scores,
                                                          double* __restrict__ stats, int* __restrict__ err) {
  const int id = blockIdx.x * 128 + threadIdx.x;
  const int T = A.compute_fp ? A.num_thresholds : 1;
  if (id >= A.num_frames * T) return;
  const int f = id / T, t = id - f * T;
  const int g0 = A.gt_off[f], n_gt = A.gt_off[f + 1] - g0, d0 = A.dt_off[f], n_det = A.dt_off[f + 1] - d0;
  if (n_det > MAXDET) {
    atomicOr(err, 1);
    return;
  }
  const double* ov = A.overlaps + A.ov_off[f];
  const double* dt = A.dt_data + (size_t)d0 * 6;
  const double* gt = A.gt_data + (size_t)g0 * 5;
  const int* ig = A.ignored_gt + g0;
  const int* id_ = A.ignored_det + d0;
  const double thresh = A.compute_fp ? A.thresholds[t] : 0.0;
  unsigned assigned[MAXDET / 32], below[MAXDET / 32];
#pragma unroll
  for (int w = 0; w < MAXDET / 32; ++w) assigned[w] = below[w] = 0u;
  if (A.compute_fp)
    for (int j = 0; j < n_det; ++j)
      if (dt[j * 6 + 5] < thresh) set(below, j);
  const double NONE = -10000000.0;
  int tp = 0, fp = 0, fn = 0;
  double sim = 0.0;
  for (int i = 0; i < n_gt; ++i) {
    if (!A.compute_fp) tp_scores[g0 + i] = __longlong_as_double(0x7FF8000000000000ll);  // NaN: no true positive in this slot
    if (ig[i] == -1) continue;
    int pick = -1;
    double best = NONE, max_ov = 0.0;
    bool picked_ignored = false;
    for (int j = 0; j < n_det; ++j) {
      if (id_[j] == -1 || bit(assigned, j) || bit(below, j)) continue;
      const double o = ov[(size_t)j * n_gt + i];
      const double sc = dt[j * 6 + 5];
      if (!A.compute_fp) {
        if (o > A.min_overlap && sc > best) { pick = j; best = sc; }
      } else if (o > A.min_overlap && (o > max_ov || picked_ignored) && id_[j] == 0) {
        max_ov = o; pick = j; best = 1.0; picked_ignored = false;
      } else if (o > A.min_overlap && best == NONE && id_[j] == 1) {
        pick = j; best = 1.0; picked_ignored = true;
      }
    }
    if (best == NONE) {
      if (ig[i] == 0) ++fn;
    } else if (ig[i] == 1 || id_[pick] == 1) {
      set(assigned, pick);
    } else {
      ++tp;
      if (!A.compute_fp) tp_scores[g0 + i] = dt[pick * 6 + 5];
      if (A.compute_aos) sim += (1.0 + cos(gt[i * 5 + 4] - dt[pick * 6 + 4])) / 2.0;
      set(assigned, pick);
    }
  }
  if (!A.compute_fp) return;
  for (int j = 0; j < n_det; ++j)
    if (!(bit(assigned, j) || id_[j] == -1 || id_[j] == 1 || bit(below, j))) ++fp;
  if (A.metric == 0) {
    const int c0 = A.dc_off[f], n_dc = A.dc_off[f + 1] - c0;
    int stuff = 0;
    for (int i = 0; i < n_dc; ++i) {
      const double* dc = A.dc_boxes + (size_t)(c0 + i) * 4;
      for (int j = 0; j < n_det; ++j) {
        if (bit(assigned, j) || id_[j] == -1 || id_[j] == 1 || bit(below, j)) continue;
        // image_box_overlap(det, dontcare, criterion 0): intersection over the detection's area (eval.py:282-312)
        const double* b = dt + j * 6;
        const double iw = fmin(b[2], dc[2]) - fmax(b[0], dc[0]);
        const double ih = fmin(b[3], dc[3]) - fmax(b[1], dc[1]);
        double o = 0.0;
        if (iw > 0 && ih > 0) o = iw * ih / ((b[2] - b[0]) * (b[3] - b[1]));
        if (o > A.min_overlap) {
          set(assigned, j);
          ++stuff;
        }
      }
    }
    fp -= stuff;
  }
  double* o = stats + ((size_t)f * T + t) * 4;
  o[0] = tp; o[1] = fp; o[2] = fn;
  o[3] = A.compute_aos ? ((tp > 0 || fp > 0) ? sim : -1.0) : 0.0;
}

// pr[t][c] = sum over the frames in order (similarity -1 = "no detections in this frame": skipped, eval.py:166-168)
__global__ void reduce_kernel(const double* __restrict__ stats, int num_frames, int T, double* __restrict__ pr) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= T * 4) return;
  const int t = id >> 2, c = id & 3;
  double s = 0.0;
  for (int f = 0; f < num_frames; ++f) {
    const double v = stats[((size_t)f * T + t) * 4 + c];
    if (c == 3 && v == -1.0) continue;
    s += v;
  }
  pr[id] = s;
}

// get_thresholds (kitti/eval.py:18-37) on scores sorted in DESCENDING order; n_out[0] = number of thresholds written (<= num_pts)
__global__ void thresholds_kernel(const double* __restrict__ sorted_scores, int n, int num_gt, int num_pts,
                                  double* __restrict__ out, int* __restrict__ n_out) {
  if (blockIdx.x || threadIdx.x) return;
  double current = 0.0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const double left = (double)(i + 1) / num_gt;
    const double right = i < n - 1 ? (double)(i + 2) / num_gt : left;
    if ((right - current) < (current - left) && i < n - 1) continue;
    if (m < num_pts) out[m] = sorted_scores[i];
    ++m;
    current += 1.0 / (num_pts - 1.0);
  }
  n_out[0] = m < num_pts ? m : num_pts;
}

}  // namespace

extern "C" {

// One pass of compute_statistics_jit over `num_frames` frames. compute_fp == 0: tp_scores (sum n_gt) receives the score of the
// detection matched to each ground truth (NaN where none) -- the input of sessd_kitti_thresholds after a descending sort;
// compute_fp != 0: stats (num_frames, num_thresholds, 4) = tp, fp, fn, similarity per frame and score threshold.
// err_flag |= 1 if a frame has more than 256 detections (not evaluated).
int sessd_kitti_statistics(const double* overlaps, const long long* ov_off, const int* gt_off, const int* dt_off, const int* dc_off,
                           const double* gt_data, const double* dt_data, const int* ignored_gt, const int* ignored_det,
                           const double* dc_boxes, int num_frames, int metric, double min_overlap, const double* thresholds,
                           int num_thresholds, int compute_fp, int compute_aos, double* tp_scores, double* stats, int* err_flag,
                           hipStream_t stream) {
  if (num_frames < 0 || (compute_fp && (num_thresholds <= 0 || !thresholds || !stats)) || (!compute_fp && !tp_scores))
    return SESSD_EINVAL;
  if (num_frames == 0) return SESSD_OK;
  EvalArgs A{overlaps, ov_off, gt_off, dt_off, dc_off, gt_data, dt_data, ignored_gt, ignored_det, dc_boxes, thresholds,
             num_frames, num_thresholds, metric, compute_fp ? 1 : 0, compute_aos ? 1 : 0, min_overlap};
  const int total = num_frames * (compute_fp ? num_thresholds : 1);
  SESSD_LAUNCH(statistics_kernel, dim3(sessd_divup(total, 128)), dim3(128), 0, stream, A, tp_scores, stats, err_flag);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_kitti_reduce(const double* stats, int num_frames, int num_thresholds, double* pr, hipStream_t stream) {
  if (num_frames < 0 || num_thresholds <= 0) return SESSD_EINVAL;
  SESSD_LAUNCH(reduce_kernel, dim3(sessd_divup(num_thresholds * 4, 64)), dim3(64), 0, stream, stats, num_frames, num_thresholds, pr);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_kitti_thresholds(const double* sorted_scores_desc, int num_scores, int num_gt, int num_sample_pts, double* thresholds,
                           int* num_thresholds, hipStream_t stream) {
  if (num_scores < 0 || num_gt <= 0 || num_sample_pts < 2) return SESSD_EINVAL;
  SESSD_LAUNCH(thresholds_kernel, dim3(1), dim3(1), 0, stream, sorted_scores_desc, num_scores, num_gt, num_sample_pts, thresholds,
               num_thresholds);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
