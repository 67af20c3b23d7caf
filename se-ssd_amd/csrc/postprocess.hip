// MultiGroupHead.predict on the device, end to end, with no host round trip:
//   det3d/models/bbox_heads/mg_head_sessd.py:893-943 (predict) and :945-1057 (get_task_detections)
//   det3d/core/bbox/box_torch_ops.py:81-147 (second_box_decode), :527-548 (rotate_nms: topk, keep[:post])
//   det3d/ops/nms/nms_cpu.py:40-51 + det3d/ops/nms/nms_cpu.h:72-168 (rotate_nms_cc, CPU/boost in the reference)
//   det3d/core/bbox/geometry.py:215-277 (frustum test), mg_head_sessd.py:1035-1045 (direction fix, range mask)
// The reference leaves the GPU three times per frame here (boolean-mask indexing, NMS on the CPU,
// frustum test in numba). Pipeline of this file, all on one stream:
//   K1 score_filter   1 thread / BEV location: sigmoid(cls) >= thresh -> 64-bit key (~score | anchor) appended
//                     with a wave-aggregated atomic (order does not matter: the key is a total order)
//   K2 topk_decode    1 workgroup / frame: running top-`pre_max` by bitonic sort in LDS (2048 keys at a time),
//                     then decode ONLY the survivors (box, rectified score, direction label, BEV corners, AABB)
//   K3 rnms_pairs / rnms_clip   suppression bitmask: AABB prefilter (float32, iou_jit eps=0) of every row over the later
//                     candidates, the surviving pairs COMPACTED into one list, then convex polygon clipping in float64 on
//                     dense lanes only; suppress when IoU >= thresh
//   K4 nms_reduce     greedy walk over the mask (staged in LDS), stops at post_max, fused with
//   K5 finalize       frustum (float64 planes), direction fix, centre-range mask, ordered compaction, and the frame's
//                     fixed-size detection record (sessd_predict_fused; the unfused kernels serve pre_max > ~1280)
// K1 can also run inside the producer of the head tensor (sessd_ssfa_fuse_head_keys, dense_conv.hip): the keys then arrive
// with the head and the frame has one launch and one 35 k-thread pass less.
// Head tensor layout consumed here: planar (B, 22, H*W): ch 0..13 box codes (anchor-major, 7 each),
// 14..15 cls, 16..19 dir (2 per anchor), 20..21 iou; anchor id = pixel*2 + a (mg_head_sessd.py:409-481).
#include "geom.hpp"

namespace {

constexpr int APL = 2;        // anchors per location
constexpr int HEAD_CH = 22;
constexpr int SORT_N = 2048;  // keys sorted per pass in LDS
constexpr int SORT_NT = 1024;

struct PostCfg {
  int num_pix;         // H*W
  float score_thresh;  // 0.3
  int pre_max, post_max;
  float nms_thresh;
  float range[6];      // post_center_range
  float dir_offset;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void score_filter_kernel(const float* __restrict__ head, PostCfg C,
                                                            unsigned long long* __restrict__ keys, int key_cap,
                                                            int* __restrict__ count) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= C.num_pix) return;
  const float* hb = head + (size_t)b * HEAD_CH * C.num_pix;
#pragma unroll
  for (int a = 0; a < APL; ++a) {
    const float s = sigmoidf_(hb[(size_t)(14 + a) * C.num_pix + pix]);
    if (s >= C.score_thresh) {
      // IoU rectification (mg_head_sessd.py:971-972): s *= ((iou + 1) * 0.5)^4
      const float r = (hb[(size_t)(20 + a) * C.num_pix + pix] + 1.0f) * 0.5f;
      const float sc = s * (r * r * r * r);
      const unsigned aid = (unsigned)(pix * APL + a);
      const unsigned long long key = ((unsigned long long)(~__float_as_uint(sc)) << 32) | aid;
      const int slot = atomicAdd(&count[b], 1);
      if (slot < key_cap) keys[(size_t)b * key_cap + slot] = key;
    }
  }
}

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* s, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (n >> 1); t += SORT_NT) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned long long a = s[lo], b = s[hi];
        if ((a > b) == up) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// Outputs per frame (stride pre_max): cand_box (7), cand_score, cand_dir, corners (8), standup (4), n_top
__global__ __launch_bounds__(SORT_NT) void topk_decode_kernel(const float* __restrict__ head,
                                                               const float* __restrict__ anchors, int anchors_per_frame,
                                                               PostCfg C, const unsigned long long* __restrict__ keys,
                                                               int key_cap, const int* __restrict__ count,
                                                               float* __restrict__ cand_box, float* __restrict__ cand_score,
                                                               int* __restrict__ cand_dir, float* __restrict__ corners,
                                                               float* __restrict__ standup, int* __restrict__ n_top,
                                                               int* __restrict__ rec_cursor, int* __restrict__ rec_base, int batch,
                                                               int* __restrict__ pair_count) {
  __shared__ unsigned long long s[SORT_N];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) pair_count[b] = 0;  // the suppression-mask launches that follow append this frame's pairs
  // detection records (sessd_predict_fused): this batch takes the ring slots cursor .. cursor + batch - 1; the last kernel of
  // the call reads the base from the workspace
  if (rec_cursor && b == 0 && threadIdx.x == 0) {
    const int c0 = *rec_cursor;
    *rec_base = c0;
    *rec_cursor = c0 + batch;
  }
  const int n = min(count[b], key_cap);
  const unsigned long long* kb = keys + (size_t)b * key_cap;
  int T = 0, pos = 0;
  while (pos < n) {
    const int take = min(n - pos, SORT_N - T);
    int npad = 64;
    while (npad < T + take) npad <<= 1;
    for (int t = threadIdx.x; t < npad - T; t += SORT_NT) s[T + t] = t < take ? kb[pos + t] : ~0ull;
    __syncthreads();
    bitonic_sort_lds(s, npad);
    T = min(T + take, C.pre_max);
    pos += take;
    __syncthreads();
  }
  if (threadIdx.x == 0) n_top[b] = T;
  const float* hb = head + (size_t)b * HEAD_CH * C.num_pix;
  for (int k = threadIdx.x; k < T; k += SORT_NT) {
    const unsigned long long key = s[k];
    const unsigned aid = (unsigned)(key & 0xFFFFFFFFull);
    const float sc = __uint_as_float(~(unsigned)(key >> 32));
    const int pix = aid / APL, a = aid % APL;
    const float* an = anchors + ((size_t)(anchors_per_frame ? (size_t)b * anchors_per_frame : 0) + aid) * 7;
    float t[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) t[q] = hb[(size_t)(a * 7 + q) * C.num_pix + pix];
    // second_box_decode (box_torch_ops.py:112-146)
    const float xa = an[0], ya = an[1], za = an[2], wa = an[3], la = an[4], ha = an[5], ra = an[6];
    const float diag = sqrtf(la * la + wa * wa);
    float bx[7];
    bx[0] = t[0] * diag + xa;
    bx[1] = t[1] * diag + ya;
    bx[2] = t[2] * ha + za;
    bx[3] = expf(t[3]) * wa;
    bx[4] = expf(t[4]) * la;
    bx[5] = expf(t[5]) * ha;
    bx[6] = t[6] + ra;
    const size_t o = (size_t)b * C.pre_max + k;
#pragma unroll
    for (int q = 0; q < 7; ++q) cand_box[o * 7 + q] = bx[q];
    cand_score[o] = sc;
    const float d0 = hb[(size_t)(16 + a * 2) * C.num_pix + pix], d1 = hb[(size_t)(17 + a * 2) * C.num_pix + pix];
    cand_dir[o] = d1 > d0 ? 1 : 0;  // torch.max: first maximum on ties
    // boxes_for_nms = box[:, [0,1,3,4,6]] -> corners (box_np_ops.py:512-532) and stand-up box
    const float det[5] = {bx[0], bx[1], bx[3], bx[4], bx[6]};
    float c8[8];
    sessd_box2d_corners(det, c8);
    float x0 = c8[0], y0 = c8[1], x1 = c8[0], y1 = c8[1];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      x0 = fminf(x0, c8[2 * q]); x1 = fmaxf(x1, c8[2 * q]);
      y0 = fminf(y0, c8[2 * q + 1]); y1 = fmaxf(y1, c8[2 * q + 1]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) corners[o * 8 + q] = c8[q];
    standup[o * 4 + 0] = x0; standup[o * 4 + 1] = y0; standup[o * 4 + 2] = x1; standup[o * 4 + 3] = y1;
  }
}

// iou_jit(eps=0) prefilter (box_np_ops.py:1007-1045), float32: stand-up IoU > 0
__device__ __forceinline__ bool rnms_prefilter(const float* si, const float* sj) {
  const float iw = fminf(si[2], sj[2]) - fmaxf(si[0], sj[0]);
  if (!(iw > 0.f)) return false;
  const float ih = fminf(si[3], sj[3]) - fmaxf(si[1], sj[1]);
  if (!(ih > 0.f)) return false;
  const float ua = (si[2] - si[0]) * (si[3] - si[1]) + (sj[2] - sj[0]) * (sj[3] - sj[1]) - iw * ih;
  const float siou = iw * ih / ua;
  return !(siou <= 0.f);
}

// polygon IoU >= thresh (nms_cpu.h:130-160: area(P & Q) / area(P | Q)), float64 clipping
__device__ __forceinline__ bool rnms_polygon(const float* ci, const float* cj, float thresh) {
  const double inter = sessd_quad_inter_area_green(ci, cj);
  if (inter <= 0) return false;
  double px[4], py[4], qx[4], qy[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { px[q] = ci[2 * q]; py[q] = ci[2 * q + 1]; qx[q] = cj[2 * q]; qy[q] = cj[2 * q + 1]; }
  const double uni = fabs(sessd_poly_area2(px, py, 4)) * 0.5 + fabs(sessd_poly_area2(qx, qy, 4)) * 0.5 - inter;
  const double ov = uni > 0 ? inter / uni : 0.0;
  return ov >= (double)thresh;
}

// The suppression mask in two balanced launches. Round 2 gave every (row, 64-column word) a wave and let the lanes that passed
// the stand-up prefilter run the float64 polygon clipping (~1500 instructions, ~6 us per wave) while the rest of the wave
// idled: nearly every one of the 8000 waves paid a full clipping round for 1-3 useful lanes (21.7 us per frame, max 38). A
// first round-3 form (one wave per row: sweep, compact in LDS, clip the row's survivors on dense lanes) was 10.7 us on typical
// frames but 68 us on frames where a few boxes overlap everything (the synthetic weights decode some boxes to kilometres):
// such a row is 16 clipping rounds on ONE wave. So:
//   rnms_pairs_kernel  one wave per row i: clears the row's mask words, sweeps the later candidates j > i with the prefilter
//                      (16-byte coalesced loads) and appends the surviving (i, j) pairs to ONE list per frame (one atomic per
//                      64 candidates); the list holds every pair up to 1024 candidates (measured: frames where kilometre-sized
//                      boxes overlap everything reach hundreds of thousands of pairs); beyond, pairs that do not fit (more than
//                      64 per row on average) are clipped on the spot;
//   rnms_clip_kernel   the list's pairs dealt out over all lanes of the launch: one clipping round on dense lanes whatever
//                      the distribution over rows; bits set with atomicOr.
constexpr int RN_ROWS = 4;       // waves (rows) per workgroup of the pair kernel
constexpr int RN_MAXN = 4096;    // candidates (rotate_nms_common's limit; pairs are packed i << 16 | j)
constexpr int RN_PAIRS_PER_ROW = 64;
// capacity of a frame's pair list: every pair up to 1024 candidates (2 MB), an average of 64 per row beyond
__host__ __device__ inline int rn_pair_cap(int n) { return n <= 1024 ? (n > 1 ? n * (n - 1) / 2 : 1) : n * RN_PAIRS_PER_ROW; }

__device__ __forceinline__ void rnms_clip_pair(const float* __restrict__ cb, int i, int j, float thresh,
                                               unsigned long long* __restrict__ mrow_base, int words) {
  const float4* c4 = reinterpret_cast<const float4*>(cb + (size_t)i * 8);
  const float4 a4 = c4[0], b4 = c4[1];
  const float4* d4 = reinterpret_cast<const float4*>(cb + (size_t)j * 8);
  const float4 e4 = d4[0], f4 = d4[1];
  const float ci[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
  const float cj[8] = {e4.x, e4.y, e4.z, e4.w, f4.x, f4.y, f4.z, f4.w};
  if (rnms_polygon(ci, cj, thresh)) atomicOr(mrow_base + (size_t)i * words + (j >> 6), 1ull << (j & 63));
}

__global__ __launch_bounds__(RN_ROWS * 64) void rnms_pairs_kernel(const int* __restrict__ n_top, int pre_max, float thresh,
                                                                   const float* __restrict__ corners,
                                                                   const float* __restrict__ standup,
                                                                   unsigned long long* __restrict__ mask, int words,
                                                                   unsigned* __restrict__ pairs, int pair_cap,
                                                                   int* __restrict__ pair_count) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * RN_ROWS + (int)(threadIdx.x >> 6);
  const int n = min(n_top[b], pre_max);
  if (i >= n) return;
  unsigned long long* mb = mask + (size_t)b * pre_max * words;
  for (int w = (i >> 6) + lane; w < words; w += 64) mb[(size_t)i * words + w] = 0ull;  // words left of the diagonal are never read
  __builtin_amdgcn_s_waitcnt(0);  // the clears precede any atomicOr of the overflow path below
  const float* cb = corners + (size_t)b * pre_max * 8;
  const float4* sb = reinterpret_cast<const float4*>(standup + (size_t)b * pre_max * 4);
  const float4 s4 = sb[i];
  const float si[4] = {s4.x, s4.y, s4.z, s4.w};
  unsigned* pl = pairs + (size_t)b * pair_cap;
  // sweep: lane c keeps the ballot of chunk c (<= 64 chunks of 64 candidates); ONE atomic per row reserves the row's run of the
  // list (a first version reserved per chunk: sixteen dependent atomic round trips per wave, 22 us on busy frames)
  const int c0 = (i + 1) >> 6;
  unsigned long long mine = 0ull;
  for (int j0 = c0 << 6; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    bool pass = false;
    if (j > i && j < n) {
      const float4 t4 = sb[j];
      const float sj[4] = {t4.x, t4.y, t4.z, t4.w};
      pass = rnms_prefilter(si, sj);
    }
    const unsigned long long bal = __ballot(pass);
    if (lane == (j0 >> 6) - c0) mine = bal;
  }
  const int cnt = __popcll(mine);
  int incl = cnt;  // inclusive prefix of the chunk counts over the lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  if (total == 0) return;
  int base = 0;
  if (lane == 0) base = atomicAdd(&pair_count[b], total);
  base = __builtin_amdgcn_readfirstlane(base);
  const int excl = incl - cnt;
  const int nchunk = ((n - 1) >> 6) - c0 + 1;
  for (int c = 0; c < nchunk; ++c) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)mine, c), hi = __builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), c);
    const unsigned long long bal = ((unsigned long long)hi << 32) | lo;
    if (bal == 0) continue;
    const int cbase = base + __builtin_amdgcn_readlane(excl, c);
    if ((bal >> lane) & 1ull) {
      const int j = ((c0 + c) << 6) + lane;
      const int idx = cbase + __popcll(bal & ((1ull << lane) - 1ull));
      if (idx < pair_cap) pl[idx] = ((unsigned)i << 16) | (unsigned)j;
      else rnms_clip_pair(cb, i, j, thresh, mb, words);  // list full: this pair is clipped here
    }
  }
}

__global__ __launch_bounds__(256) void rnms_clip_kernel(int pre_max, float thresh, const float* __restrict__ corners,
                                                         unsigned long long* __restrict__ mask, int words,
                                                         const unsigned* __restrict__ pairs, int pair_cap,
                                                         const int* __restrict__ pair_count) {
  const int b = blockIdx.y;
  const int np = min(pair_count[b], pair_cap);
  const float* cb = corners + (size_t)b * pre_max * 8;
  unsigned long long* mb = mask + (size_t)b * pre_max * words;
  const unsigned* pl = pairs + (size_t)b * pair_cap;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < np; p += gridDim.x * 256) {
    const unsigned pr = pl[p];
    rnms_clip_pair(cb, (int)(pr >> 16), (int)(pr & 0xFFFFu), thresh, mb, words);
  }
}

// both launches; pair_count[b] must be zero (sessd_predict_fused: cleared by topk_decode_kernel; rotate_nms_common: by its set kernel)
int launch_rnms_mask(const int* n_top, int batch, int pre_max, float thresh, const float* corners, const float* standup,
                     unsigned long long* mask, int words, unsigned* pairs, int* pair_count, hipStream_t stream) {
  if (pre_max > RN_MAXN) return SESSD_EINVAL;  // pairs are packed i << 16 | j
  const int pair_cap = rn_pair_cap(pre_max);
  SESSD_LAUNCH(rnms_pairs_kernel, dim3(sessd_divup(pre_max, RN_ROWS), batch), dim3(RN_ROWS * 64), 0, stream, n_top, pre_max,
               thresh, corners, standup, mask, words, pairs, pair_cap, pair_count);
  SESSD_CHECK_LAUNCH();
  const int g = sessd_divup(pair_cap, 256) < 512 ? sessd_divup(pair_cap, 256) : 512;
  SESSD_LAUNCH(rnms_clip_kernel, dim3(g, batch), dim3(256), 0, stream, pre_max, thresh, corners, mask, words, pairs, pair_cap,
               pair_count);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// Post-NMS filters of one frame by ONE wave (mg_head_sessd.py:1024-1055): frustum (B,1,6,4,3) float64 surfaces (or null),
// direction fix, centre-range mask, ordered compaction. keep = the frame's kept candidate rows (global or LDS), nk of them.
// rec != nullptr: the finalized rows also go to rec[row * 9 + {box 7, score, label}] (LDS staging of the detection record).
// Returns the number of detections written (wave-uniform).
__device__ __forceinline__ int finalize_wave(const PostCfg& C, int b, int lane, const int* keep, int nk,
                                             const float* __restrict__ cand_box, const float* __restrict__ cand_score,
                                             const int* __restrict__ cand_dir, const double* __restrict__ frustum,
                                             float* __restrict__ out_box, float* __restrict__ out_score,
                                             int* __restrict__ out_label, float* rec) {
  double nx[6], ny[6], nz[6], nd[6];
  if (frustum) {
    const double* f = frustum + (size_t)b * 72;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double* s = f + k * 12;
      const double ax = s[0] - s[3], ay = s[1] - s[4], az = s[2] - s[5];
      const double bx = s[3] - s[6], by = s[4] - s[7], bz = s[5] - s[8];
      nx[k] = ay * bz - az * by;
      ny[k] = az * bx - ax * bz;
      nz[k] = ax * by - ay * bx;
      nd[k] = -s[0] * nx[k] - s[1] * ny[k] - s[2] * nz[k];
    }
  }
  int written = 0;
  for (int base = 0; base < nk; base += 64) {
    const int k = base + lane;
    bool ok = k < nk;
    float bx[7] = {0, 0, 0, 0, 0, 0, 0};
    float sc = 0.f;
    if (ok) {
      const size_t o = (size_t)b * C.pre_max + keep[k];
#pragma unroll
      for (int q = 0; q < 7; ++q) bx[q] = cand_box[o * 7 + q];
      sc = cand_score[o];
      if (frustum) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          const double sign = (double)bx[0] * nx[p] + (double)bx[1] * ny[p] + (double)bx[2] * nz[p] + nd[p];
          if (sign >= 0) ok = false;
        }
      }
      const bool opp = ((bx[6] - C.dir_offset) > 0.f) != (cand_dir[o] == 1);
      if (opp) bx[6] += 3.14159265358979323846f;
      ok = ok && bx[0] >= C.range[0] && bx[1] >= C.range[1] && bx[2] >= C.range[2] && bx[0] <= C.range[3] &&
           bx[1] <= C.range[4] && bx[2] <= C.range[5];
    }
    const unsigned long long bal = __ballot(ok);
    if (ok) {
      const int dst = written + __popcll(bal & ((1ull << lane) - 1ull));
      const size_t o = (size_t)b * C.post_max + dst;
#pragma unroll
      for (int q = 0; q < 7; ++q) out_box[o * 7 + q] = bx[q];
      out_score[o] = sc;
      out_label[o] = 0;
      if (rec) {
#pragma unroll
        for (int q = 0; q < 7; ++q) rec[dst * 9 + q] = bx[q];
        rec[dst * 9 + 7] = sc;
        rec[dst * 9 + 8] = 0.f;
      }
    }
    written += __popcll(bal);
  }
  return written;
}

__global__ __launch_bounds__(64) void finalize_kernel(PostCfg C, const int* __restrict__ keep, const int* __restrict__ n_keep,
                                                       const float* __restrict__ cand_box, const float* __restrict__ cand_score,
                                                       const int* __restrict__ cand_dir, const double* __restrict__ frustum,
                                                       float* __restrict__ out_box, float* __restrict__ out_score,
                                                       int* __restrict__ out_label, int* __restrict__ out_count) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nk = min(n_keep[b], C.post_max);
  const int written = finalize_wave(C, b, lane, keep + (size_t)b * C.post_max, nk, cand_box, cand_score, cand_dir, frustum, out_box,
                                    out_score, out_label, nullptr);
  if (lane == 0) out_count[b] = written;
}

// The greedy walk of the rotated NMS over a suppression mask in LDS by ONE wave (nms_cpu.h:86-168: candidates in score order,
// a candidate is kept unless a kept one suppresses it; stops at post_max). lane w owns word w of the running "removed" set.
// Round 2 tested every candidate of a block in turn (~20 scalar-ish instructions x 1000 candidates = 10 us of the 22 us this
// kernel took); here the next kept candidate of a block is the lowest clear bit of (~removed & valid) -- s_ff1 -- so the loop
// runs once per KEPT candidate (<= post_max in total). Kept rows go to `keep_out` (LDS or global); returns their number.
__device__ __forceinline__ int nms_walk_lds(const unsigned long long* sm, int n, int words, int post_max, int lane, int* keep_out) {
  auto rdlane64 = [](unsigned long long v, int l) -> unsigned long long {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
  };
  const int cb = sessd_divup(n, 64);
  unsigned long long removed = 0;
  int nk = 0;
  for (int blk = 0; blk < cb && nk < post_max; ++blk) {
    const int row = blk * 64 + lane;
    const unsigned long long diag = row < n ? sm[(size_t)row * words + blk] : 0ull;
    unsigned long long rem = rdlane64(removed, blk);
    const int lim = min(64, n - blk * 64);
    const unsigned long long valid = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
    unsigned long long kept = 0;
    unsigned long long avail = ~rem & valid;
    while (avail && nk < post_max) {
      const int bb = __builtin_amdgcn_readfirstlane(__builtin_ctzll(avail));
      kept |= 1ull << bb;
      ++nk;
      rem |= rdlane64(diag, bb) | (1ull << bb);
      avail = ~rem & valid;
    }
    {  // kept rows of this block -> output list, ascending
      const int base = nk - __popcll(kept);
      unsigned long long k2 = kept;
      for (int t = 0; k2; ++t, k2 &= k2 - 1)
        if (lane == 0) keep_out[base + t] = blk * 64 + __builtin_ctzll(k2);
    }
    if (nk >= post_max) break;
    if (lane > blk && lane < cb) {
      unsigned long long acc = 0;
      for (unsigned long long k2 = kept; k2; k2 &= k2 - 1)
        acc |= sm[(size_t)(blk * 64 + __builtin_ctzll(k2)) * words + lane];
      removed |= acc;
    }
  }
  return nk;
}

__device__ __forceinline__ void stage_mask_lds(unsigned long long* sm, const unsigned long long* mb, int n, int words) {
  // blind 16-byte copy of the n x words matrix (words left of the diagonal were never written by the mask kernel and are
  // never read by the walk); 8 independent loads per thread in flight
  const int total2 = (n * words + 1) >> 1;
  const uint4* src = reinterpret_cast<const uint4*>(mb);
  uint4* dst = reinterpret_cast<uint4*>(sm);
#pragma unroll 8
  for (int idx = threadIdx.x; idx < total2; idx += 1024) dst[idx] = src[idx];
}

// The greedy reduction with the WHOLE suppression mask staged in LDS (pre_max*ceil(pre_max/64)*8 B = 128 KB for pre_max 1000;
// 160 KB LDS per CU: 1024 threads copy it in one coalesced sweep, then one wave walks it at LDS latency instead of one
// dependent global load per kept row),
// FUSED with the post-NMS filters and the frame's detection record: after the walk the kept rows sit in LDS,
// wave 0 runs finalize_wave on them (<= post_max boxes), the finalized rows are staged in LDS and all threads write the
// fixed-size record (post_max x 9 floats + count; rows beyond the count zero) -- three launches of round 2 (nms_reduce 22 us,
// finalize 4.9 us, pack_detections 4.2 us) as one. Dynamic LDS: mask | keep[post_max] | rec[post_max * 9].
// records == nullptr: no record. rec_base = the slot counter value of this batch's first frame (written by topk_decode).
__global__ __launch_bounds__(1024) void nms_reduce_finalize_kernel(PostCfg C, const int* __restrict__ n_top,
                                                                   const unsigned long long* __restrict__ mask, int words,
                                                                   const float* __restrict__ cand_box,
                                                                   const float* __restrict__ cand_score,
                                                                   const int* __restrict__ cand_dir,
                                                                   const double* __restrict__ frustum, float* __restrict__ out_box,
                                                                   float* __restrict__ out_score, int* __restrict__ out_label,
                                                                   int* __restrict__ out_count, float* __restrict__ records,
                                                                   int* __restrict__ rec_count, int capacity,
                                                                   const int* __restrict__ rec_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
  __shared__ int s_written;
  const int b = blockIdx.x;
  const int n = min(n_top[b], C.pre_max);
  const size_t mask_words = (size_t)C.pre_max * words;
  int* s_keep = reinterpret_cast<int*>(sm + mask_words);
  float* s_rec = reinterpret_cast<float*>(s_keep + C.post_max);
  stage_mask_lds(sm, mask + (size_t)b * mask_words, n, words);
  __syncthreads();
  if (threadIdx.x < 64) {
    const int nk = nms_walk_lds(sm, n, words, C.post_max, threadIdx.x, s_keep);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // lane 0's list is read by the other lanes of this wave
    const int written = finalize_wave(C, b, threadIdx.x, s_keep, nk, cand_box, cand_score, cand_dir, frustum, out_box, out_score,
                                      out_label, records ? s_rec : nullptr);
    if (threadIdx.x == 0) {
      out_count[b] = written;
      s_written = written;
    }
  }
  if (!records) return;
  __syncthreads();
  const int written = s_written;
  const int slot = (rec_base[0] + b) % capacity;
  float* dst = records + (size_t)slot * C.post_max * 9;
  for (int e = threadIdx.x; e < C.post_max * 9; e += 1024) dst[e] = e < written * 9 ? s_rec[e] : 0.f;
  if (threadIdx.x == 0) rec_count[slot] = written;
}

// ---- detection records for the end-of-job gather (tools/dist_test.py:150-186 gathers pickled per-rank dicts; here every frame
// leaves one fixed-size record on the device: (post_max, 9) float32 [box 7 | score | label] + a count), appended by the frame's
// own launch sequence so that a captured graph needs no host-side bookkeeping: slot = (*cursor + b) % capacity, cursor += batch.
__global__ __launch_bounds__(256) void pack_detections_kernel(const float* __restrict__ box, const float* __restrict__ score,
                                                               const int* __restrict__ label, const int* __restrict__ count,
                                                               int batch, int post_max, float* __restrict__ records,
                                                               int* __restrict__ rec_count, int capacity, int* __restrict__ cursor,
                                                               const int* __restrict__ base_in) {
  // base_in: the batch's first slot was already taken (and the cursor advanced) by topk_decode_kernel
  const int c0 = base_in ? *base_in : *cursor;
  const int total = batch * post_max * 9;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / (post_max * 9), r = (e - b * post_max * 9) / 9, q = e % 9;
    const int n = count[b];
    float v = 0.f;
    if (r < n) v = q < 7 ? box[((size_t)b * post_max + r) * 7 + q] : (q == 7 ? score[(size_t)b * post_max + r] : (float)label[(size_t)b * post_max + r]);
    records[((size_t)((c0 + b) % capacity) * post_max + r) * 9 + q] = v;
  }
  if (threadIdx.x < batch) rec_count[(c0 + threadIdx.x) % capacity] = count[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0 && !base_in) *cursor = c0 + batch;
}

struct PostWs {
  unsigned long long* keys;
  int* count;
  float* cand_box;
  float* cand_score;
  int* cand_dir;
  float* corners;
  float* standup;
  int* n_top;
  unsigned long long* mask;
  int* keep;
  int* n_keep;
  int* rec_base;
  unsigned* pairs;
  int* pair_count;
};

size_t post_ws_layout(int batch, int num_anchors, int pre_max, int post_max, PostWs* w, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = sessd_align(off + bytes, 256);
    return o;
  };
  const int words = sessd_divup(pre_max, 64);
  size_t o_keys = take((size_t)batch * num_anchors * 8);
  size_t o_count = take((size_t)batch * 4);
  size_t o_box = take((size_t)batch * pre_max * 7 * 4);
  size_t o_score = take((size_t)batch * pre_max * 4);
  size_t o_dir = take((size_t)batch * pre_max * 4);
  size_t o_cor = take((size_t)batch * pre_max * 8 * 4);
  size_t o_su = take((size_t)batch * pre_max * 4 * 4);
  size_t o_nt = take((size_t)batch * 4);
  size_t o_mask = take((size_t)batch * pre_max * words * 8);
  size_t o_keep = take((size_t)batch * post_max * 4);
  size_t o_nk = take((size_t)batch * 4);
  size_t o_rb = take(4);
  size_t o_pairs = take((size_t)batch * rn_pair_cap(pre_max) * 4);
  size_t o_pc = take((size_t)batch * 4);
  if (w) {
    w->keys = (unsigned long long*)(base + o_keys);
    w->count = (int*)(base + o_count);
    w->cand_box = (float*)(base + o_box);
    w->cand_score = (float*)(base + o_score);
    w->cand_dir = (int*)(base + o_dir);
    w->corners = (float*)(base + o_cor);
    w->standup = (float*)(base + o_su);
    w->n_top = (int*)(base + o_nt);
    w->mask = (unsigned long long*)(base + o_mask);
    w->keep = (int*)(base + o_keep);
    w->n_keep = (int*)(base + o_nk);
    w->rec_base = (int*)(base + o_rb);
    w->pairs = (unsigned*)(base + o_pairs);
    w->pair_count = (int*)(base + o_pc);
  }
  return off;
}

// batched variant of the greedy reduction (one wave per frame)
__global__ __launch_bounds__(64) void nms_reduce_batch_kernel(const int* __restrict__ n_top, int pre_max,
                                                               const unsigned long long* __restrict__ mask, int words,
                                                               int post_max, int* __restrict__ keep, int* __restrict__ n_keep) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = min(n_top[b], pre_max);
  const unsigned long long* mb = mask + (size_t)b * pre_max * words;
  int* kb = keep + (size_t)b * post_max;
  const int cb = sessd_divup(n, 64);
  unsigned long long removed = 0;  // lane w owns word w (pre_max <= 4096)
  int nk = 0;
  for (int blk = 0; blk < cb && nk < post_max; ++blk) {
    const int row = blk * 64 + lane;
    unsigned long long diag = row < n ? mb[(size_t)row * words + blk] : 0ull;
    unsigned long long rem = __shfl(removed, blk, 64);
    unsigned long long kept = 0;
    const int lim = min(64, n - blk * 64);
    for (int bb = 0; bb < lim; ++bb) {
      const unsigned long long d = __shfl(diag, bb, 64);
      if (!((rem >> bb) & 1ull) && nk < post_max) {
        kept |= 1ull << bb;
        if (lane == 0) kb[nk] = blk * 64 + bb;
        ++nk;
        rem |= d;
      }
    }
    if (nk >= post_max) break;
    for (int bb = 0; bb < lim; ++bb) {
      if (!((kept >> bb) & 1ull)) continue;
      if (lane > blk && lane < cb) removed |= mb[(size_t)(blk * 64 + bb) * words + lane];
    }
  }
  if (lane == 0) n_keep[b] = nk;
}

}  // namespace

extern "C" {

size_t sessd_predict_workspace_bytes(int batch, int num_anchors, int pre_max_size, int post_max_size) {
  return post_ws_layout(batch, num_anchors, pre_max_size, post_max_size, nullptr, nullptr);
}

// head (B,22,H*W) planar, anchors (A,7) shared by all frames (anchors_per_frame = 0) or (B,A,7),
// frustum (B,1,6,4,3) float64 or NULL. Outputs: out_box (B,post,7), out_score (B,post), out_label (B,post) int32,
// out_count (B,) -- rows [0,out_count[b]) are the detections of frame b in NMS order.
// ext_keys / ext_key_count (both or neither): the score-filter keys (B, 2 * num_pixels) uint64 and their per-frame counts were
// already produced with the head tensor (sessd_ssfa_fuse_head_keys; counts zeroed by the caller before that launch) -- the
// count clear and the score_filter launch are skipped.
// records != NULL: every frame also leaves its fixed-size detection record (sessd_pack_detections' layout and ring rule:
// slot = (*cursor + b) % capacity_frames, *cursor += batch) from inside the last launch.
// 3 launches per call with external keys and pre_max_size such that the suppression mask fits the LDS (<= ~1280): top-k +
// decode, suppression mask, greedy walk + filters + record.
int sessd_predict_fused(const float* head, int batch, int num_pixels, const float* anchors, int anchors_per_frame,
                        const double* frustum, float score_thresh, int pre_max_size, int post_max_size, float nms_iou_thresh,
                        const float* post_center_range6, float direction_offset, float* out_box, float* out_score,
                        int* out_label, int* out_count, const unsigned long long* ext_keys, const int* ext_key_count,
                        float* records, int* record_counts, int capacity_frames, int* cursor, void* workspace,
                        size_t workspace_bytes, hipStream_t stream) {
  if (batch < 1 || num_pixels < 1 || pre_max_size < 1 || pre_max_size > 4096 || post_max_size < 1) return SESSD_EINVAL;
  if (pre_max_size > SORT_N - 64) return SESSD_EINVAL;  // running top-k keeps pre_max + a fresh chunk in 2048 slots
  if ((ext_keys == nullptr) != (ext_key_count == nullptr)) return SESSD_EINVAL;
  if (records && (!record_counts || !cursor || capacity_frames < batch || batch > 256)) return SESSD_EINVAL;
  const int A = num_pixels * APL;
  PostWs w;
  if (post_ws_layout(batch, A, pre_max_size, post_max_size, &w, (char*)workspace) > workspace_bytes)
    return SESSD_EWORKSPACE;
  PostCfg C;
  C.num_pix = num_pixels;
  C.score_thresh = score_thresh;
  C.pre_max = pre_max_size;
  C.post_max = post_max_size;
  C.nms_thresh = nms_iou_thresh;
  for (int i = 0; i < 6; ++i) C.range[i] = post_center_range6[i];
  C.dir_offset = direction_offset;
  const unsigned long long* keys = ext_keys;
  const int* key_count = ext_key_count;
  if (!ext_keys) {
    SESSD_FILL(w.count, 0, batch, stream);
    SESSD_LAUNCH(score_filter_kernel, dim3(sessd_divup(num_pixels, 256), batch), dim3(256), 0, stream, head, C,
                       w.keys, A, w.count);
    SESSD_CHECK_LAUNCH();
    keys = w.keys;
    key_count = w.count;
  }
  SESSD_LAUNCH(topk_decode_kernel, dim3(batch), dim3(SORT_NT), 0, stream, head, anchors, anchors_per_frame, C,
                     keys, A, key_count, w.cand_box, w.cand_score, w.cand_dir, w.corners, w.standup, w.n_top,
                     records ? cursor : (int*)nullptr, w.rec_base, batch, w.pair_count);
  SESSD_CHECK_LAUNCH();
  const int words = sessd_divup(pre_max_size, 64);
  {
    const int rc = launch_rnms_mask(w.n_top, batch, pre_max_size, nms_iou_thresh, w.corners, w.standup, w.mask, words, w.pairs,
                                    w.pair_count, stream);
    if (rc != SESSD_OK) return rc;
  }
  const size_t lds = (size_t)pre_max_size * words * 8 + (size_t)post_max_size * 4 + (size_t)post_max_size * 9 * 4;
  if (lds <= 160 * 1024 - 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      SESSD_TRY(hipFuncSetAttribute((const void*)nms_reduce_finalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 1024));
      attr_set = true;
    }
    SESSD_LAUNCH(nms_reduce_finalize_kernel, dim3(batch), dim3(1024), lds, stream, C, w.n_top, w.mask, words, w.cand_box,
                       w.cand_score, w.cand_dir, frustum, out_box, out_score, out_label, out_count, records, record_counts,
                       capacity_frames, w.rec_base);
    SESSD_CHECK_LAUNCH();
    return SESSD_OK;
  }
  SESSD_LAUNCH(nms_reduce_batch_kernel, dim3(batch), dim3(64), 0, stream, w.n_top, pre_max_size, w.mask, words,
                     post_max_size, w.keep, w.n_keep);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(finalize_kernel, dim3(batch), dim3(64), 0, stream, C, w.keep, w.n_keep, w.cand_box, w.cand_score,
                     w.cand_dir, frustum, out_box, out_score, out_label, out_count);
  SESSD_CHECK_LAUNCH();
  if (records) {
    SESSD_LAUNCH(pack_detections_kernel, dim3(1), dim3(256), 0, stream, out_box, out_score, out_label, out_count, batch,
                 post_max_size, records, record_counts, capacity_frames, (int*)nullptr, w.rec_base);
    SESSD_CHECK_LAUNCH();
  }
  return SESSD_OK;
}

int sessd_predict(const float* head, int batch, int num_pixels, const float* anchors, int anchors_per_frame,
                  const double* frustum, float score_thresh, int pre_max_size, int post_max_size, float nms_iou_thresh,
                  const float* post_center_range6, float direction_offset, float* out_box, float* out_score,
                  int* out_label, int* out_count, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return sessd_predict_fused(head, batch, num_pixels, anchors, anchors_per_frame, frustum, score_thresh, pre_max_size,
                             post_max_size, nms_iou_thresh, post_center_range6, direction_offset, out_box, out_score, out_label,
                             out_count, nullptr, nullptr, nullptr, nullptr, 0, nullptr, workspace, workspace_bytes, stream);
}

// Stand-alone rotated NMS with the predict-path semantics (box_torch_ops.rotate_nms after its topk):
// dets (N,5) [x,y,w,l,r] sorted by descending score. keep (device int32[post_max]), num_keep (device int).
size_t sessd_rotate_nms_workspace_bytes(int num_boxes) {
  const int words = sessd_divup(num_boxes > 0 ? num_boxes : 1, 64);
  return sessd_align((size_t)num_boxes * 12 * 4, 256) + sessd_align((size_t)num_boxes * words * 8, 256) + 512 +
         sessd_align((size_t)rn_pair_cap(num_boxes > 0 ? num_boxes : 1) * 4, 256);
}

}  // extern "C"

namespace {
__global__ __launch_bounds__(256) void rnms_prep_kernel(const float* __restrict__ dets, int n, float* __restrict__ corners,
                                                         float* __restrict__ standup) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float c8[8];
  sessd_box2d_corners(dets + (size_t)i * 5, c8);
  float x0 = c8[0], y0 = c8[1], x1 = c8[0], y1 = c8[1];
#pragma unroll
  for (int q = 1; q < 4; ++q) {
    x0 = fminf(x0, c8[2 * q]); x1 = fmaxf(x1, c8[2 * q]);
    y0 = fminf(y0, c8[2 * q + 1]); y1 = fmaxf(y1, c8[2 * q + 1]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) corners[(size_t)i * 8 + q] = c8[q];
  standup[(size_t)i * 4 + 0] = x0; standup[(size_t)i * 4 + 1] = y0; standup[(size_t)i * 4 + 2] = x1; standup[(size_t)i * 4 + 3] = y1;
}
__global__ void set_int_kernel(int* p, int v) { p[0] = v; p[1] = 0; }  // n_top, pair_count
// corners given by the caller (det3d.ops.nms.nms.rotate_non_max_suppression_cpu, nms_cpu.h:72-168): copy + AABB
__global__ __launch_bounds__(256) void rnms_prep_corners_kernel(const float* __restrict__ in_corners, int n,
                                                                 float* __restrict__ corners, float* __restrict__ standup) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float c8[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) c8[q] = in_corners[(size_t)i * 8 + q];
  float x0 = c8[0], y0 = c8[1], x1 = c8[0], y1 = c8[1];
#pragma unroll
  for (int q = 1; q < 4; ++q) {
    x0 = fminf(x0, c8[2 * q]); x1 = fmaxf(x1, c8[2 * q]);
    y0 = fminf(y0, c8[2 * q + 1]); y1 = fmaxf(y1, c8[2 * q + 1]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) corners[(size_t)i * 8 + q] = c8[q];
  standup[(size_t)i * 4 + 0] = x0; standup[(size_t)i * 4 + 1] = y0; standup[(size_t)i * 4 + 2] = x1; standup[(size_t)i * 4 + 3] = y1;
}
}  // namespace

static int rotate_nms_common(const float* dets, const float* in_corners, int num_boxes, float iou_thresh, int post_max_size,
                             int* keep, int* num_keep, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (num_boxes < 0 || num_boxes > 4096 || post_max_size < 1) return SESSD_EINVAL;
  if (workspace_bytes < sessd_rotate_nms_workspace_bytes(num_boxes)) return SESSD_EWORKSPACE;
  if (num_boxes == 0) {
    SESSD_FILL(num_keep, 0, 1, stream);
    return SESSD_OK;
  }
  char* base = (char*)workspace;
  float* corners = (float*)base;
  float* standup = corners + (size_t)num_boxes * 8;
  size_t off = sessd_align((size_t)num_boxes * 12 * 4, 256);
  const int words = sessd_divup(num_boxes, 64);
  unsigned long long* mask = (unsigned long long*)(base + off);
  off += sessd_align((size_t)num_boxes * words * 8, 256);
  int* n_top = (int*)(base + off);
  unsigned* pairs = (unsigned*)(base + off + 512);
  SESSD_LAUNCH(set_int_kernel, dim3(1), dim3(1), 0, stream, n_top, num_boxes);
  if (in_corners)
    SESSD_LAUNCH(rnms_prep_corners_kernel, dim3(sessd_divup(num_boxes, 256)), dim3(256), 0, stream, in_corners,
                       num_boxes, corners, standup);
  else
    SESSD_LAUNCH(rnms_prep_kernel, dim3(sessd_divup(num_boxes, 256)), dim3(256), 0, stream, dets, num_boxes, corners,
                       standup);
  SESSD_CHECK_LAUNCH();
  {
    const int rc = launch_rnms_mask(n_top, 1, num_boxes, iou_thresh, corners, standup, mask, words, pairs, n_top + 1, stream);
    if (rc != SESSD_OK) return rc;
  }
  SESSD_LAUNCH(nms_reduce_batch_kernel, dim3(1), dim3(64), 0, stream, n_top, num_boxes, mask, words, post_max_size,
                     keep, num_keep);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// The same greedy rotated NMS on caller-supplied corner quads (N,4,2), already in descending-score order: the device
// form of det3d.ops.nms.nms.rotate_non_max_suppression_cpu (nms_cpu.h:72-168; the stand-up IoU prefilter is recomputed
// from the corners' bounding boxes = what nms_cpu.py:45-49 passes in).
extern "C" int sessd_rotate_nms_corners_sorted(const float* corners, int num_boxes, float iou_thresh, int post_max_size,
                                               int* keep, int* num_keep, void* workspace, size_t workspace_bytes,
                                               hipStream_t stream) {
  if (!corners && num_boxes > 0) return SESSD_EINVAL;
  return rotate_nms_common(nullptr, corners, num_boxes, iou_thresh, post_max_size, keep, num_keep, workspace, workspace_bytes,
                           stream);
}

extern "C" int sessd_rotate_nms_sorted(const float* dets, int num_boxes, float iou_thresh, int post_max_size, int* keep,
                                       int* num_keep, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return rotate_nms_common(dets, nullptr, num_boxes, iou_thresh, post_max_size, keep, num_keep, workspace, workspace_bytes,
                           stream);
}

// ---- spconv.utils.rbbox_iou / rbbox_intersection (spconv v1 box_iou: boost polygons on the host), as imported by
// det3d/core/bbox/box_np_ops.py:9 and used by riou_cc / rinter_cc :20-50: pairwise IoU / intersection area of convex quads
// given as corners, skipped (0) where the caller's stand-up IoU is <= standup_thresh. Same float64 clipper as the rotated NMS.
namespace {
__global__ __launch_bounds__(256) void quads_pairwise_kernel(int mode, const float* __restrict__ ca, int n,
                                                              const float* __restrict__ cb, int k,
                                                              const float* __restrict__ standup_iou, float standup_thresh,
                                                              float* __restrict__ out) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (size_t)n * k) return;
  const int i = (int)(id / k), j = (int)(id - (size_t)i * k);
  float v = 0.f;
  if (standup_iou[id] > standup_thresh) {
    const float* pi = ca + (size_t)i * 8;
    const float* pj = cb + (size_t)j * 8;
    const double inter = sessd_quad_inter_area_green(pi, pj);
    if (inter > 0) {
      if (mode == 0) {
        double px[4], py[4], qx[4], qy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { px[q] = pi[2 * q]; py[q] = pi[2 * q + 1]; qx[q] = pj[2 * q]; qy[q] = pj[2 * q + 1]; }
        const double uni = fabs(sessd_poly_area2(px, py, 4)) * 0.5 + fabs(sessd_poly_area2(qx, qy, 4)) * 0.5 - inter;
        v = uni > 0 ? (float)(inter / uni) : 0.f;
      } else {
        v = (float)inter;
      }
    }
  }
  out[id] = v;
}
}  // namespace

// mode 0: IoU, 1: intersection area. corners (n,4,2) / (k,4,2) float32, standup_iou and out (n,k) row-major.
extern "C" int sessd_quads_pairwise(int mode, const float* corners_a, int n, const float* corners_b, int k,
                                    const float* standup_iou, float standup_thresh, float* out, hipStream_t stream) {
  if (n < 0 || k < 0 || mode < 0 || mode > 1) return SESSD_EINVAL;
  if (n == 0 || k == 0) return SESSD_OK;
  const size_t total = (size_t)n * k;
  SESSD_LAUNCH(quads_pairwise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mode, corners_a, n,
               corners_b, k, standup_iou, standup_thresh, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

extern "C" int sessd_pack_detections(const float* out_box, const float* out_score, const int* out_label, const int* out_count,
                                     int batch, int post_max_size, float* records, int* record_counts, int capacity_frames,
                                     int* cursor, hipStream_t stream) {
  if (batch <= 0 || batch > 256 || post_max_size <= 0 || capacity_frames < batch) return SESSD_EINVAL;
  SESSD_LAUNCH(pack_detections_kernel, dim3(1), dim3(256), 0, stream, out_box, out_score, out_label, out_count, batch,
               post_max_size, records, record_counts, capacity_frames, cursor, (const int*)nullptr);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}
