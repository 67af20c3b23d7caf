// Rotated BEV / 3-D IoU and bitmask NMS on gfx950: the device side of the reference's
// `iou3d_cuda` torch extension
//   det3d/core/iou3d/src/iou3d.cpp:24-262,270-281   (entry points)
//   det3d/core/iou3d/src/iou3d_kernel.cu:270-470    (7 kernels)
// One thread per box pair (the geometry is branchy scalar code, not a contraction), tiles of
// 16x16 pairs per 256-thread workgroup; the 32 rectangles of a tile are rotated ONCE into LDS
// instead of once per pair, and each thread's intersection polygon lives in a k-major LDS list.
// NMS: 64x64 suppression bitmask tiles (upper triangle only) + an ON-DEVICE greedy reduction by
// one wave (the reference copies the mask to the host and loops there, iou3d.cpp:117-164).
#include "geom.hpp"

namespace {

constexpr int TP = 16;  // pair tile edge

enum { MODE_OVERLAP = 0, MODE_IOU_BEV = 1, MODE_IOU_3D = 2, MODE_IOU_NORMAL = 3 };

struct BoxRow {
  float v[7];
};

template <int W>
__device__ __forceinline__ void load_rect(const float* __restrict__ b, sessd_rect& R, float& z1, float& z2) {
  if (W == 5) {
    sessd_rect_init(R, b[0], b[1], b[2], b[3], b[4]);
    z1 = 0.f; z2 = 0.f;
  } else {
    sessd_rect_init(R, b[0], b[1], b[3], b[4], b[6]);
    z1 = b[2]; z2 = b[5];
  }
}

__device__ __forceinline__ float iou_from_overlap(const sessd_rect& A, const sessd_rect& B, float so) {
  float sa = (A.x2 - A.x1) * (A.y2 - A.y1);
  float sb = (B.x2 - B.x1) * (B.y2 - B.y1);
  return so / fmaxf(sa + sb - so, SESSD_IOU_EPS);
}

template <int MODE>
__device__ __forceinline__ float pair_value(const sessd_rect& A, float az1, float az2, const sessd_rect& B, float bz1,
                                            float bz2, sessd_ptlist L) {
  if (MODE == MODE_OVERLAP) return sessd_rect_overlap_f32(A, B, L);
  if (MODE == MODE_IOU_BEV) return iou_from_overlap(A, B, sessd_rect_overlap_f32(A, B, L));
  if (MODE == MODE_IOU_NORMAL) {
    float l = fmaxf(A.x1, B.x1), r = fminf(A.x2, B.x2), t = fmaxf(A.y1, B.y1), bt = fminf(A.y2, B.y2);
    float w = fmaxf(r - l, 0.f), h = fmaxf(bt - t, 0.f), s = w * h;
    float sa = (A.x2 - A.x1) * (A.y2 - A.y1), sb = (B.x2 - B.x1) * (B.y2 - B.y1);
    return s / fmaxf(sa + sb - s, SESSD_IOU_EPS);
  }
  // MODE_IOU_3D (iou3d_kernel.cu:256-268): zero when the z ranges do not overlap
  float va = (A.x2 - A.x1) * (A.y2 - A.y1) * (az2 - az1);
  float vb = (B.x2 - B.x1) * (B.y2 - B.y1) * (bz2 - bz1);
  float dh = fmaxf(fminf(az2, bz2) - fmaxf(az1, bz1), SESSD_IOU_EPS);
  if (dh == SESSD_IOU_EPS) return 0.f;
  float vo = sessd_rect_overlap_f32(A, B, L) * dh;
  return vo / fmaxf(va + vb - vo, SESSD_IOU_EPS);
}

struct TileLds {
  sessd_rect ra[TP];
  sessd_rect rb[TP];
  float za[TP][2];
  float zb[TP][2];
  float px[SESSD_IOU_MAXPTS][TP * TP];
  float py[SESSD_IOU_MAXPTS][TP * TP];
  float pa[SESSD_IOU_MAXPTS][TP * TP];
};

template <int MODE, int W>
__global__ __launch_bounds__(TP* TP) void pairwise_kernel(int na, const float* __restrict__ a, int nb,
                                                           const float* __restrict__ b, float* __restrict__ out) {
  __shared__ TileLds S;
  const int tid = threadIdx.y * TP + threadIdx.x;
  const int a0 = blockIdx.y * TP, b0 = blockIdx.x * TP;
  if (tid < TP) {
    int i = a0 + tid;
    if (i < na) load_rect<W>(a + (size_t)i * W, S.ra[tid], S.za[tid][0], S.za[tid][1]);
  } else if (tid < 2 * TP) {
    int j = b0 + tid - TP;
    if (j < nb) load_rect<W>(b + (size_t)j * W, S.rb[tid - TP], S.zb[tid - TP][0], S.zb[tid - TP][1]);
  }
  __syncthreads();
  const int i = a0 + threadIdx.y, j = b0 + threadIdx.x;
  if (i >= na || j >= nb) return;
  sessd_ptlist L = {&S.px[0][tid], &S.py[0][tid], &S.pa[0][tid], TP * TP};
  out[(size_t)i * nb + j] = pair_value<MODE>(S.ra[threadIdx.y], S.za[threadIdx.y][0], S.za[threadIdx.y][1],
                                             S.rb[threadIdx.x], S.zb[threadIdx.x][0], S.zb[threadIdx.x][1], L);
}

struct AlignedLds {
  float px[SESSD_IOU_MAXPTS][256];
  float py[SESSD_IOU_MAXPTS][256];
  float pa[SESSD_IOU_MAXPTS][256];
};

// boxes_aligned_overlap_kernel (iou3d_kernel.cu:284-293): pair (i,i)
__global__ __launch_bounds__(256) void aligned_overlap_kernel(int n, const float* __restrict__ a,
                                                               const float* __restrict__ b, float* __restrict__ out) {
  __shared__ AlignedLds S;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  sessd_rect A, B;
  float z1, z2;
  load_rect<5>(a + (size_t)i * 5, A, z1, z2);
  load_rect<5>(b + (size_t)i * 5, B, z1, z2);
  sessd_ptlist L = {&S.px[0][threadIdx.x], &S.py[0][threadIdx.x], &S.pa[0][threadIdx.x], 256};
  out[i] = sessd_rect_overlap_f32(A, B, L);
}

// ---- NMS ---------------------------------------------------------------------------------
// mask[i][cb] bit t set <=> box (cb*64+t) is suppressed by box i (only t > i within the diagonal
// block, iou3d_kernel.cu:345-349). One 64-thread workgroup per (row block, col block >= row block).
struct NmsLds {
  sessd_rect rb[64];
  float zb[64][2];
  float px[SESSD_IOU_MAXPTS][64];
  float py[SESSD_IOU_MAXPTS][64];
  float pa[SESSD_IOU_MAXPTS][64];
};

template <int MODE, int W>
__global__ __launch_bounds__(64) void nms_mask_kernel(int n, float thresh, const float* __restrict__ boxes,
                                                       unsigned long long* __restrict__ mask) {
  const int rblk = blockIdx.y, cblk = blockIdx.x;
  if (cblk < rblk) return;  // never read by the reduction
  __shared__ NmsLds S;
  const int t = threadIdx.x;
  const int ncol = min(n - cblk * 64, 64);
  const int nrow = min(n - rblk * 64, 64);
  if (t < ncol) load_rect<W>(boxes + (size_t)(cblk * 64 + t) * W, S.rb[t], S.zb[t][0], S.zb[t][1]);
  __syncthreads();
  if (t >= nrow) return;
  const int i = rblk * 64 + t;
  sessd_rect A;
  float az1, az2;
  load_rect<W>(boxes + (size_t)i * W, A, az1, az2);
  sessd_ptlist L = {&S.px[0][t], &S.py[0][t], &S.pa[0][t], 64};
  unsigned long long bits = 0;
  const int start = (rblk == cblk) ? t + 1 : 0;
  for (int k = start; k < ncol; ++k) {
    float v = pair_value<MODE>(A, az1, az2, S.rb[k], S.zb[k][0], S.zb[k][1], L);
    if (v > thresh) bits |= 1ull << k;
  }
  const int cb = sessd_divup(n, 64);
  mask[(size_t)i * cb + cblk] = bits;
}

}  // namespace

// Greedy reduction of a suppression bitmask by ONE wave, shared with nms.hip.
// Lane w owns the removed-bits word(s) w, w+64, ... . Blocks of 64 candidate rows are resolved
// in registers (64 readlane steps on the diagonal word), then the surviving rows are OR-ed in.
// keep[] receives the kept row numbers (ascending), *num_keep their count; stops at max_keep.
__global__ __launch_bounds__(64) void sessd_nms_reduce_kernel(const int* __restrict__ n_dev, int n_host,
                                                               const unsigned long long* __restrict__ mask,
                                                               int mask_stride_words, int max_keep,
                                                               long long* __restrict__ keep64, int* __restrict__ keep32,
                                                               int* __restrict__ num_keep) {
  const int lane = threadIdx.x;
  int n = n_dev ? n_dev[0] : n_host;
  if (n > n_host) n = n_host;
  const int cb = sessd_divup(n, 64);
  // up to 64*SESSD_NMS_WPL column blocks (n <= 64*64*WPL)
  constexpr int WPL = 4;
  unsigned long long removed[WPL];
#pragma unroll
  for (int q = 0; q < WPL; ++q) removed[q] = 0;
  int nk = 0;
  for (int blk = 0; blk < cb && nk < max_keep; ++blk) {
    const int row = blk * 64 + lane;
    const bool valid = row < n;
    // diagonal word of my row, and the removed word of this block (owned by lane blk%64, slot blk/64)
    unsigned long long diag = valid ? mask[(size_t)row * mask_stride_words + blk] : 0ull;
    unsigned long long rem = 0;
#pragma unroll
    for (int q = 0; q < WPL; ++q) {
      unsigned long long r = __shfl(removed[q], blk & 63, 64);
      if ((blk >> 6) == q) rem = r;
    }
    unsigned long long kept = 0;
    const int lim = min(64, n - blk * 64);
    for (int b = 0; b < lim; ++b) {
      unsigned long long d = __shfl(diag, b, 64);
      if (!((rem >> b) & 1ull)) {
        if (nk < max_keep) {
          kept |= 1ull << b;
          if (lane == 0) {
            if (keep64) keep64[nk] = blk * 64 + b;
            if (keep32) keep32[nk] = blk * 64 + b;
          }
          ++nk;
          rem |= d;
        }
      }
    }
    if (nk >= max_keep) break;
    // OR the kept rows into every later removed word: lane w handles words w, w+64, ...
    for (int b = 0; b < lim; ++b) {
      if (!((kept >> b) & 1ull)) continue;  // wave-uniform
      const unsigned long long* rowp = mask + (size_t)(blk * 64 + b) * mask_stride_words;
#pragma unroll
      for (int q = 0; q < WPL; ++q) {
        int w = q * 64 + lane;
        if (w > blk && w < cb) removed[q] |= rowp[w];
      }
    }
  }
  if (lane == 0) *num_keep = nk;
}

extern "C" {

// mode: 0 overlap area (N,5)x(M,5); 1 BEV IoU (N,5)x(M,5); 2 3-D IoU (N,7)x(M,7)
int sessd_boxes_pairwise(int mode, const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* out,
                         hipStream_t stream) {
  if (num_a < 0 || num_b < 0) return SESSD_EINVAL;
  if (num_a == 0 || num_b == 0) return SESSD_OK;
  dim3 grid(sessd_divup(num_b, TP), sessd_divup(num_a, TP)), block(TP, TP);
  switch (mode) {
    case MODE_OVERLAP:
      hipLaunchKernelGGL((pairwise_kernel<MODE_OVERLAP, 5>), grid, block, 0, stream, num_a, boxes_a, num_b, boxes_b, out);
      break;
    case MODE_IOU_BEV:
      hipLaunchKernelGGL((pairwise_kernel<MODE_IOU_BEV, 5>), grid, block, 0, stream, num_a, boxes_a, num_b, boxes_b, out);
      break;
    case MODE_IOU_3D:
      hipLaunchKernelGGL((pairwise_kernel<MODE_IOU_3D, 7>), grid, block, 0, stream, num_a, boxes_a, num_b, boxes_b, out);
      break;
    default:
      return SESSD_EINVAL;
  }
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_boxes_aligned_overlap_bev(const float* boxes_a, const float* boxes_b, int num, float* out,
                                    hipStream_t stream) {
  if (num < 0) return SESSD_EINVAL;
  if (num == 0) return SESSD_OK;
  hipLaunchKernelGGL(aligned_overlap_kernel, dim3(sessd_divup(num, 256)), dim3(256), 0, stream, num, boxes_a, boxes_b,
                     out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

size_t sessd_nms_workspace_bytes(int num_boxes) {
  return sessd_align((size_t)num_boxes * sessd_divup(num_boxes > 0 ? num_boxes : 1, 64) * 8 + 8, 256);
}

// mode: 0 rotated BEV (N,5) | 1 3-D (N,7) | 2 axis aligned (N,5). Boxes sorted by descending score.
// keep (device, int64[num_boxes]) and num_keep (device int) are written on `stream`; no host sync.
int sessd_nms_sorted(int mode, const float* boxes, int num_boxes, float thresh, long long* keep, int* num_keep,
                     void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (num_boxes < 0 || mode < 0 || mode > 2) return SESSD_EINVAL;
  if (num_boxes > 64 * 64 * 4) return SESSD_EINVAL;
  if (workspace_bytes < sessd_nms_workspace_bytes(num_boxes)) return SESSD_EWORKSPACE;
  if (num_boxes == 0) {
    SESSD_FILL(num_keep, 0, 1, stream);
    return SESSD_OK;
  }
  unsigned long long* mask = (unsigned long long*)workspace;
  const int cb = sessd_divup(num_boxes, 64);
  dim3 grid(cb, cb), block(64);
  if (mode == 0)
    hipLaunchKernelGGL((nms_mask_kernel<MODE_IOU_BEV, 5>), grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  else if (mode == 1)
    hipLaunchKernelGGL((nms_mask_kernel<MODE_IOU_3D, 7>), grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  else
    hipLaunchKernelGGL((nms_mask_kernel<MODE_IOU_NORMAL, 5>), grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  SESSD_CHECK_LAUNCH();
  hipLaunchKernelGGL(sessd_nms_reduce_kernel, dim3(1), dim3(64), 0, stream, (const int*)nullptr, num_boxes, mask, cb,
                     num_boxes, keep, (int*)nullptr, num_keep);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
