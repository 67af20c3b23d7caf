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

enum { MODE_OVERLAP = 0, MODE_IOU_BEV = 1, MODE_IOU_3D = 2, MODE_IOU_NORMAL = 3, MODE_IOU_3D_CPU = 4 };

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
  // the _cpu twin (iou3d_cpu.cpp:306-336) has no early return: disjoint z ranges give overlap * 1e-8, not exactly 0
  if (MODE != MODE_IOU_3D_CPU && dh == SESSD_IOU_EPS) return 0.f;
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

// ---- numba-convention rotated IoU (det3d/ops/nms/nms_gpu.py:183-419,580-633) ----------------------------
// boxes [cx, cy, w, l, angle]; corners rotated clockwise-positive about the centre, intersection polygon =
// corners of each quad inside the other (closed dot-product test) + 16 edge/edge crossings, ordered by a
// pseudo-angle key with an insertion sort, area = fan of |triangle| areas. Per-thread scratch (8 points + keys)
// lives in k-major LDS like the iou3d polygon lists.
struct RotLds {
  float px[8][256];
  float py[8][256];
  float vs[8][256];
};

__device__ __forceinline__ void rot_corners(const float* r, float* c) {
  const float a_cos = cosf(r[4]), a_sin = sinf(r[4]);
  const float xs[4] = {-r[2] / 2, -r[2] / 2, r[2] / 2, r[2] / 2}, ys[4] = {-r[3] / 2, r[3] / 2, r[3] / 2, -r[3] / 2};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c[2 * i] = a_cos * xs[i] + a_sin * ys[i] + r[0];
    c[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + r[1];
  }
}

__device__ __forceinline__ bool rot_in_quad(float x, float y, const float* c) {
  const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
  const float ap0 = x - c[0], ap1 = y - c[1];
  const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

__device__ __forceinline__ bool rot_seg_inter(const float* p1, const float* p2, int i, int j, float* ox, float* oy) {
  const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
  const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
  const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
  const bool acd = DA1 * CA0 > CA1 * DA0;
  const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
  if (acd == bcd) return false;
  const bool abc = CA1 * BA0 > BA1 * CA0;
  const bool abd = DA1 * BA0 > BA1 * DA0;
  if (abc == abd) return false;
  const float DC0 = D0 - C0, DC1 = D1 - C1;
  const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
  const float DH = BA1 * DC0 - BA0 * DC1;
  *ox = (ABBA * DC0 - BA0 * CDDC) / DH;
  *oy = (ABBA * DC1 - BA1 * CDDC) / DH;
  return true;
}

__device__ float rot_inter(const float* r1, const float* r2, float* PX, float* PY, float* VS, int stride) {
  {  // exact zero when the bounding circles are disjoint (no corner inside, no crossing)
    const float dx = r1[0] - r2[0], dy = r1[1] - r2[1];
    const float ra = 0.5f * sqrtf(r1[2] * r1[2] + r1[3] * r1[3]), rb = 0.5f * sqrtf(r2[2] * r2[2] + r2[3] * r2[3]);
    const float reach = ra + rb + 1e-2f;
    if (dx * dx + dy * dy > reach * reach) return 0.f;
  }
  float c1[8], c2[8];
  rot_corners(r1, c1);
  rot_corners(r2, c2);
  int n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (rot_in_quad(c1[2 * i], c1[2 * i + 1], c2) && n < 8) { PX[n * stride] = c1[2 * i]; PY[n * stride] = c1[2 * i + 1]; ++n; }
    if (rot_in_quad(c2[2 * i], c2[2 * i + 1], c1) && n < 8) { PX[n * stride] = c2[2 * i]; PY[n * stride] = c2[2 * i + 1]; ++n; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x, y;
      if (rot_seg_inter(c1, c2, i, j, &x, &y) && n < 8) { PX[n * stride] = x; PY[n * stride] = y; ++n; }
    }
  if (n < 3) return 0.f;
  float cx = 0.f, cy = 0.f;
  for (int i = 0; i < n; ++i) { cx += PX[i * stride]; cy += PY[i * stride]; }
  cx /= n; cy /= n;
  for (int i = 0; i < n; ++i) {
    float vx = PX[i * stride] - cx, vy = PY[i * stride] - cy;
    const float d = sqrtf(vx * vx + vy * vy);
    vx = vx / d; vy = vy / d;
    if (vy < 0) vx = -2 - vx;
    VS[i * stride] = vx;
  }
  for (int i = 1; i < n; ++i) {
    if (VS[(i - 1) * stride] > VS[i * stride]) {
      const float temp = VS[i * stride], tx = PX[i * stride], ty = PY[i * stride];
      int j = i;
      while (j > 0 && VS[(j - 1) * stride] > temp) {
        VS[j * stride] = VS[(j - 1) * stride]; PX[j * stride] = PX[(j - 1) * stride]; PY[j * stride] = PY[(j - 1) * stride];
        --j;
      }
      VS[j * stride] = temp; PX[j * stride] = tx; PY[j * stride] = ty;
    }
  }
  float area = 0.f;
  const float x0 = PX[0], y0 = PY[0];
  for (int i = 0; i < n - 2; ++i) {
    const float bx = PX[(i + 1) * stride], by = PY[(i + 1) * stride], c0 = PX[(i + 2) * stride], c1y = PY[(i + 2) * stride];
    area += fabsf(((x0 - c0) * (by - c1y) - (y0 - c1y) * (bx - c0)) / 2.0f);
  }
  return area;
}

__device__ __forceinline__ float rot_iou_crit(const float* r1, const float* r2, int criterion, float* PX, float* PY,
                                              float* VS, int stride) {
  const float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
  const float it = rot_inter(r1, r2, PX, PY, VS, stride);
  if (criterion == -1) return it / (a1 + a2 - it);
  if (criterion == 0) return it / a1;
  if (criterion == 1) return it / a2;
  return it;
}

// out[n][k] = devRotateIoUEval(query[k], boxes[n], criterion) -- query FIRST, as rotate_iou_kernel_eval does
__global__ __launch_bounds__(256) void rotate_iou_eval_kernel(const float* __restrict__ boxes, int N,
                                                               const float* __restrict__ query, int K, int criterion,
                                                               float* __restrict__ out) {
  __shared__ RotLds S;
  const int k = blockIdx.x * 16 + (threadIdx.x & 15), n = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (n >= N || k >= K) return;
  float rb[5], rq[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) { rb[q] = boxes[(size_t)n * 5 + q]; rq[q] = query[(size_t)k * 5 + q]; }
  const int t = threadIdx.x;
  out[(size_t)n * K + k] = rot_iou_crit(rq, rb, criterion, &S.px[0][t], &S.py[0][t], &S.vs[0][t], 256);
}

// det3d/datasets/utils/eval.py:324-367 box3d_overlap = d3_box_overlap_kernel over rotate_iou_gpu_eval(criterion 2): the rotated
// BEV intersection area times the overlap of the height ranges, normalised by union (-1) / volume(box) (0) / volume(query) (1).
// boxes (N,7), query (K,7) rows [loc 3, dims 3, rot]; z_axis = index of the height axis among the three (KITTI camera: 1),
// z_center = where the location sits in the height (camera: 1.0 = bottom face). One launch instead of a device call for the
// intersections plus a numba loop for the rest. The BEV box of a row is what remains after dropping the height axis
// (eval.py:359-364), in the reference's [c0, c1, d0, d1, rot] order.
__global__ __launch_bounds__(256) void box3d_overlap_eval_kernel(const double* __restrict__ boxes, int N,
                                                                  const double* __restrict__ query, int K, int criterion,
                                                                  int z_axis, double z_center, double* __restrict__ out) {
  __shared__ RotLds S;
  const int k = blockIdx.x * 16 + (threadIdx.x & 15), n = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (n >= N || k >= K) return;
  const double* b = boxes + (size_t)n * 7;
  const double* q = query + (size_t)k * 7;
  // the rotated part runs in float32 like rotate_iou_gpu_eval (nms_gpu.py:655-657 casts its inputs), the height / volume part in
  // the annotation dtype (float64) like the numba loop
  float rb[5], rq[5];
  int w = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (a != z_axis) {
      rb[w] = (float)b[a]; rb[w + 2] = (float)b[a + 3];
      rq[w] = (float)q[a]; rq[w + 2] = (float)q[a + 3];
      ++w;
    }
  rb[4] = (float)b[6]; rq[4] = (float)q[6];
  const int t = threadIdx.x;
  double rinc = (double)rot_iou_crit(rq, rb, 2, &S.px[0][t], &S.py[0][t], &S.vs[0][t], 256);
  if (rinc > 0.0) {
    const double min_z = fmin(b[z_axis] + b[z_axis + 3] * (1.0 - z_center), q[z_axis] + q[z_axis + 3] * (1.0 - z_center));
    const double max_z = fmax(b[z_axis] - b[z_axis + 3] * z_center, q[z_axis] - q[z_axis + 3] * z_center);
    const double iw = min_z - max_z;
    if (iw > 0.0) {
      const double area1 = b[3] * b[4] * b[5], area2 = q[3] * q[4] * q[5];
      const double inc = iw * rinc;
      const double ua = criterion == -1 ? (area1 + area2 - inc) : (criterion == 0 ? area1 : (criterion == 1 ? area2 : 1.0));
      rinc = inc / ua;
    } else {
      rinc = 0.0;
    }
  }
  out[(size_t)n * K + k] = rinc;
}

// rotate_nms_kernel (nms_gpu.py:422-458): suppress when devRotateIoU(row, col) > thresh; boxes (N,5) sorted
__global__ __launch_bounds__(64) void rotate_nms_numba_mask_kernel(int n, float thresh, const float* __restrict__ boxes,
                                                                    unsigned long long* __restrict__ mask) {
  const int rblk = blockIdx.y, cblk = blockIdx.x;
  if (cblk < rblk) return;
  __shared__ float bb[64][5];
  __shared__ float px[8][64], py[8][64], vs[8][64];
  const int t = threadIdx.x;
  const int ncol = min(n - cblk * 64, 64), nrow = min(n - rblk * 64, 64);
  if (t < ncol)
#pragma unroll
    for (int q = 0; q < 5; ++q) bb[t][q] = boxes[(size_t)(cblk * 64 + t) * 5 + q];
  __syncthreads();
  if (t >= nrow) return;
  const int i = rblk * 64 + t;
  float ri[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) ri[q] = boxes[(size_t)i * 5 + q];
  unsigned long long bits = 0;
  for (int k = (rblk == cblk) ? t + 1 : 0; k < ncol; ++k)
    if (rot_iou_crit(ri, bb[k], -1, &px[0][t], &py[0][t], &vs[0][t], 64) > thresh) bits |= 1ull << k;
  mask[(size_t)i * sessd_divup(n, 64) + cblk] = bits;
}

// numba nms_kernel / iou_device (nms_gpu.py:22-104): axis aligned with the +1 pixel convention; boxes (N,5) x1,y1,x2,y2,score
__global__ __launch_bounds__(64) void nms_plus1_mask_kernel(int n, float thresh, const float* __restrict__ boxes,
                                                             unsigned long long* __restrict__ mask) {
  const int rblk = blockIdx.y, cblk = blockIdx.x;
  if (cblk < rblk) return;
  __shared__ float bb[64][4];
  const int t = threadIdx.x;
  const int ncol = min(n - cblk * 64, 64), nrow = min(n - rblk * 64, 64);
  if (t < ncol)
#pragma unroll
    for (int q = 0; q < 4; ++q) bb[t][q] = boxes[(size_t)(cblk * 64 + t) * 5 + q];
  __syncthreads();
  if (t >= nrow) return;
  const int i = rblk * 64 + t;
  const float a0 = boxes[(size_t)i * 5], a1 = boxes[(size_t)i * 5 + 1], a2 = boxes[(size_t)i * 5 + 2], a3 = boxes[(size_t)i * 5 + 3];
  unsigned long long bits = 0;
  for (int k = (rblk == cblk) ? t + 1 : 0; k < ncol; ++k) {
    const float left = fmaxf(a0, bb[k][0]), right = fminf(a2, bb[k][2]), top = fmaxf(a1, bb[k][1]), bottom = fminf(a3, bb[k][3]);
    const float w = fmaxf(right - left + 1, 0.f), h = fmaxf(bottom - top + 1, 0.f), inter = w * h;
    const float sa = (a2 - a0 + 1) * (a3 - a1 + 1), sb = (bb[k][2] - bb[k][0] + 1) * (bb[k][3] - bb[k][1] + 1);
    if (inter / (sa + sb - inter) > thresh) bits |= 1ull << k;
  }
  mask[(size_t)i * sessd_divup(n, 64) + cblk] = bits;
}

// det3d/ops/nms/nms_cpu.h:24-70 non_max_suppression_cpu (and nms_cpu.py:100-127 nms_jit): axis aligned, areas and
// overlaps widened by eps, suppress when IoU >= thresh; boxes (N, stride >= 4) x1,y1,x2,y2
__global__ __launch_bounds__(64) void nms_eps_mask_kernel(int n, float thresh, float eps, const float* __restrict__ boxes,
                                                           int stride, unsigned long long* __restrict__ mask) {
  const int rblk = blockIdx.y, cblk = blockIdx.x;
  if (cblk < rblk) return;
  __shared__ float bb[64][4];
  const int t = threadIdx.x;
  const int ncol = min(n - cblk * 64, 64), nrow = min(n - rblk * 64, 64);
  if (t < ncol)
#pragma unroll
    for (int q = 0; q < 4; ++q) bb[t][q] = boxes[(size_t)(cblk * 64 + t) * stride + q];
  __syncthreads();
  if (t >= nrow) return;
  const int i = rblk * 64 + t;
  const float a0 = boxes[(size_t)i * stride], a1 = boxes[(size_t)i * stride + 1], a2 = boxes[(size_t)i * stride + 2],
              a3 = boxes[(size_t)i * stride + 3];
  const float sa = (a2 - a0 + eps) * (a3 - a1 + eps);
  unsigned long long bits = 0;
  for (int k = (rblk == cblk) ? t + 1 : 0; k < ncol; ++k) {
    const float w = fminf(a2, bb[k][2]) - fmaxf(a0, bb[k][0]) + eps;
    if (w > 0.f) {
      const float hgt = fminf(a3, bb[k][3]) - fmaxf(a1, bb[k][1]) + eps;
      if (hgt > 0.f) {
        const float inter = w * hgt, sb = (bb[k][2] - bb[k][0] + eps) * (bb[k][3] - bb[k][1] + eps);
        if (inter / (sa + sb - inter) >= thresh) bits |= 1ull << k;
      }
    }
  }
  mask[(size_t)i * sessd_divup(n, 64) + cblk] = bits;
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

// mode: 0 overlap area (N,5)x(M,5); 1 BEV IoU (N,5)x(M,5); 2 3-D IoU (N,7)x(M,7); 3 3-D IoU as boxes_iou3d_cpu computes it
int sessd_boxes_pairwise(int mode, const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* out,
                         hipStream_t stream) {
  if (num_a < 0 || num_b < 0) return SESSD_EINVAL;
  if (num_a == 0 || num_b == 0) return SESSD_OK;
  dim3 grid(sessd_divup(num_b, TP), sessd_divup(num_a, TP)), block(TP, TP);
  switch (mode) {
    case MODE_OVERLAP:
      SESSD_LAUNCH((pairwise_kernel<MODE_OVERLAP, 5>), grid, block, 0, stream, num_a, boxes_a, num_b, boxes_b, out);
      break;
    case MODE_IOU_BEV:
      SESSD_LAUNCH((pairwise_kernel<MODE_IOU_BEV, 5>), grid, block, 0, stream, num_a, boxes_a, num_b, boxes_b, out);
      break;
    case MODE_IOU_3D:
      SESSD_LAUNCH((pairwise_kernel<MODE_IOU_3D, 7>), grid, block, 0, stream, num_a, boxes_a, num_b, boxes_b, out);
      break;
    case 3:  // the convention of boxes_iou3d_cpu (public mode number 3)
      SESSD_LAUNCH((pairwise_kernel<MODE_IOU_3D_CPU, 7>), grid, block, 0, stream, num_a, boxes_a, num_b, boxes_b, out);
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
  SESSD_LAUNCH(aligned_overlap_kernel, dim3(sessd_divup(num, 256)), dim3(256), 0, stream, num, boxes_a, boxes_b,
                     out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

size_t sessd_nms_workspace_bytes(int num_boxes) {
  return sessd_align((size_t)num_boxes * sessd_divup(num_boxes > 0 ? num_boxes : 1, 64) * 8 + 8, 256);
}

// mode: 0 rotated BEV (N,5) | 1 3-D (N,7) | 2 axis aligned (N,5) | 3 numba rotate_nms (N,5 [cx,cy,w,l,r]) |
// 4 numba nms (+1 pixel convention, N,5 [x1,y1,x2,y2,-]). Boxes sorted by descending score.
// keep (device, int64[num_boxes]) and num_keep (device int) are written on `stream`; no host sync.
int sessd_nms_sorted(int mode, const float* boxes, int num_boxes, float thresh, long long* keep, int* num_keep,
                     void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (num_boxes < 0 || mode < 0 || mode > 4) return SESSD_EINVAL;
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
    SESSD_LAUNCH((nms_mask_kernel<MODE_IOU_BEV, 5>), grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  else if (mode == 1)
    SESSD_LAUNCH((nms_mask_kernel<MODE_IOU_3D, 7>), grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  else if (mode == 2)
    SESSD_LAUNCH((nms_mask_kernel<MODE_IOU_NORMAL, 5>), grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  else if (mode == 3)
    SESSD_LAUNCH(rotate_nms_numba_mask_kernel, grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  else
    SESSD_LAUNCH(nms_plus1_mask_kernel, grid, block, 0, stream, num_boxes, thresh, boxes, mask);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(sessd_nms_reduce_kernel, dim3(1), dim3(64), 0, stream, (const int*)nullptr, num_boxes, mask, cb,
                     num_boxes, keep, (int*)nullptr, num_keep);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// det3d.ops.nms.nms.non_max_suppression_cpu (nms_cpu.h:24-70) on the device: boxes (N, stride >= 4) [x1,y1,x2,y2,...]
// already in descending-score order; IoU with eps-widened extents, suppress at >= thresh.
int sessd_nms_axis_eps_sorted(const float* boxes, int stride, int num_boxes, float thresh, float eps, long long* keep,
                              int* num_keep, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (num_boxes < 0 || stride < 4 || num_boxes > 64 * 64 * 4) return SESSD_EINVAL;
  if (workspace_bytes < sessd_nms_workspace_bytes(num_boxes)) return SESSD_EWORKSPACE;
  if (num_boxes == 0) {
    SESSD_FILL(num_keep, 0, 1, stream);
    return SESSD_OK;
  }
  unsigned long long* mask = (unsigned long long*)workspace;
  const int cb = sessd_divup(num_boxes, 64);
  SESSD_LAUNCH(nms_eps_mask_kernel, dim3(cb, cb), dim3(64), 0, stream, num_boxes, thresh, eps, boxes, stride, mask);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(sessd_nms_reduce_kernel, dim3(1), dim3(64), 0, stream, (const int*)nullptr, num_boxes, mask, cb,
                     num_boxes, keep, (int*)nullptr, num_keep);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// det3d/ops/nms/nms_gpu.py:636-672 rotate_iou_gpu_eval (and :541-577 rotate_iou_gpu with criterion -1):
// boxes (N,5), query (K,5) [cx,cy,w,l,angle] -> out (N,K); criterion -1 IoU | 0 inter/area(query) | 1 inter/area(box) | 2 inter
int sessd_rotate_iou_eval(const float* boxes, int num_boxes, const float* query, int num_query, int criterion, float* out,
                          hipStream_t stream) {
  if (num_boxes < 0 || num_query < 0) return SESSD_EINVAL;
  if (num_boxes == 0 || num_query == 0) return SESSD_OK;
  SESSD_LAUNCH(rotate_iou_eval_kernel, dim3(sessd_divup(num_query, 16), sessd_divup(num_boxes, 16)), dim3(256), 0, stream,
                     boxes, num_boxes, query, num_query, criterion, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// det3d/datasets/utils/eval.py:324-367 box3d_overlap on the device (see box3d_overlap_eval_kernel)
int sessd_box3d_overlap_eval(const double* boxes, int num_boxes, const double* query, int num_query, int criterion, int z_axis,
                             double z_center, double* out, hipStream_t stream) {
  if (num_boxes < 0 || num_query < 0 || z_axis < 0 || z_axis > 2 || criterion < -1 || criterion > 2) return SESSD_EINVAL;
  if (num_boxes == 0 || num_query == 0) return SESSD_OK;
  SESSD_LAUNCH(box3d_overlap_eval_kernel, dim3(sessd_divup(num_query, 16), sessd_divup(num_boxes, 16)), dim3(256), 0, stream,
               boxes, num_boxes, query, num_query, criterion, z_axis, z_center, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
