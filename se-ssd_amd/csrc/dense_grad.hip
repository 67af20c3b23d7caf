// Weight gradient of the dense BEV convolutions on gfx950 (backward of det3d/models/necks/rpn_v1.py:135-235
// inside the training step, det3d/torchie/trainer/trainer_sessd.py:250-275). The DATA gradients need no kernel of their
// own: d(conv3x3 s1) is a 3x3 s1 conv with flipped, transposed weights, d(conv3x3 s2) is the 3x3 s2 transposed conv,
// d(transposed conv) is the 3x3 s2 conv, d(1x1) is a 1x1 -- all launches of dense_conv.hip with re-packed weights.
//
//   dW[co][ci][ky][kx] = sum over (b, y, x) of  gout[b][co][y][x] * inp[b][ci][y*S + ky - P][x*S + kx - P]
//
// A GEMM with the output PIXELS as the reduction axis, on v_mfma_f32_32x32x2_f32:
//   D_tap[co 32][ci 32] += A[co][pixel 2] * B_tap[pixel 2][ci]
//   A lane (i, h): 4 consecutive pixels x0+4h .. x0+4h+3 of gout row y, channel co0+i  (one aligned 16-byte load);
//                  MFMA step c uses component c, i.e. the k index of the MFMA is the pixel pair {x0+c, x0+4+c}
//   B lane (j, h): the same 4 pixels of inp channel ci0+j shifted by the tap: per input row one aligned 16-byte load plus
//                  the left / right neighbour (stride 1) or two 16-byte loads plus the left neighbour (stride 2); rows or
//                  columns outside the image get an out-of-range buffer offset (the hardware returns 0 = zero padding)
// One wave owns a 32x32 (co, ci) tile with all taps (9 accumulators) over one chunk of output rows; the <= 64 chunk
// partials are summed in order by a second kernel: deterministic, no float atomics.
#include <cstdlib>
#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ld1(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ f32x4 ld4(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
#define SESSD_OOB 0x80000000u

struct WgArgs {
  const float* inp;   // (B, ci, hi, wi)
  const float* gout;  // (B, co, ho, wo)
  float* partial;     // [nchunks][co][ci][KS*KS]
  int batch, ci, hi, wi, co, ho, wo;
  int rows_per_chunk, cib_n;
};

template <int KS, int S>
__global__ __launch_bounds__(64) void conv_wgrad_partial_kernel(WgArgs A) {
  constexpr int NT = KS * KS, P = KS / 2;
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int cob = blockIdx.x / A.cib_n, cib = blockIdx.x - cob * A.cib_n;
  const int co = cob * 32 + i, ci = cib * 32 + i;
  const bool co_ok = co < A.co, ci_ok = ci < A.ci;
  const int R = A.batch * A.ho;
  const int r0 = blockIdx.y * A.rows_per_chunk, r1 = min(R, r0 + A.rows_per_chunk);
  const rsrc_t gr = make_rsrc(A.gout, (unsigned)((size_t)A.batch * A.co * A.ho * A.wo * 4));
  const rsrc_t xr = make_rsrc(A.inp, (unsigned)((size_t)A.batch * A.ci * A.hi * A.wi * 4));
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  for (int r = r0; r < r1; ++r) {
    const int b = r / A.ho, y = r - b * A.ho;
    const unsigned grow = co_ok ? (unsigned)((((size_t)b * A.co + co) * A.ho + y) * A.wo * 4) : SESSD_OOB;
    unsigned xrow[KS];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int yin = y * S + ky - P;
      xrow[ky] = (ci_ok && yin >= 0 && yin < A.hi) ? (unsigned)((((size_t)b * A.ci + ci) * A.hi + yin) * A.wi * 4) : SESSD_OOB;
    }
    // operands of the NEXT 8-pixel step are fetched before the 4 x NT MFMAs of the current one (two register sets, the loop is
    // unrolled by two so that the set index is static; a step past the end of the row loads out-of-range offsets = zeros)
#define SESSD_WG_LOAD(SET, X0)                                                                                      \
  {                                                                                                                 \
    const int xa = (X0) + 4 * h;                                                                                    \
    const bool in = (X0) < A.wo;                                                                                    \
    a[SET] = ld4(gr, (grow == SESSD_OOB || !in) ? SESSD_OOB : grow + (unsigned)xa * 4u);                            \
    _Pragma("unroll") for (int ky = 0; ky < KS; ++ky) {                                                             \
      const unsigned row = in ? xrow[ky] : SESSD_OOB;                                                               \
      if (KS == 1) {                                                                                                \
        bt[SET][0] = ld4(xr, row == SESSD_OOB ? SESSD_OOB : row + (unsigned)xa * 4u);                               \
      } else if (S == 1) {                                                                                          \
        const f32x4 c = ld4(xr, row == SESSD_OOB ? SESSD_OOB : row + (unsigned)xa * 4u);                            \
        const float l = ld1(xr, (row == SESSD_OOB || xa == 0) ? SESSD_OOB : row + (unsigned)(xa - 1) * 4u);         \
        const float rr = ld1(xr, (row == SESSD_OOB || xa + 4 >= A.wi) ? SESSD_OOB : row + (unsigned)(xa + 4) * 4u); \
        bt[SET][ky * 3 + 0] = (f32x4){l, c.x, c.y, c.z};                                                            \
        bt[SET][ky * 3 + 1] = c;                                                                                    \
        bt[SET][ky * 3 + 2] = (f32x4){c.y, c.z, c.w, rr};                                                           \
      } else {                                                                                                      \
        const int xi = 2 * xa; /* input column of output pixel xa at kx = 1 */                                      \
        const f32x4 v0 = ld4(xr, row == SESSD_OOB ? SESSD_OOB : row + (unsigned)xi * 4u);                           \
        const f32x4 v1 = ld4(xr, row == SESSD_OOB ? SESSD_OOB : row + (unsigned)(xi + 4) * 4u);                     \
        const float l = ld1(xr, (row == SESSD_OOB || xi == 0) ? SESSD_OOB : row + (unsigned)(xi - 1) * 4u);         \
        bt[SET][ky * 3 + 0] = (f32x4){l, v0.y, v0.w, v1.y};                                                         \
        bt[SET][ky * 3 + 1] = (f32x4){v0.x, v0.z, v1.x, v1.z};                                                      \
        bt[SET][ky * 3 + 2] = (f32x4){v0.y, v0.w, v1.y, v1.w};                                                      \
      }                                                                                                             \
    }                                                                                                               \
  }
#define SESSD_WG_MMA(SET)                                                                                           \
  _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                     \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                  \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[SET][c], bt[SET][t][c], acc[t], 0, 0, 0);
    f32x4 a[2];
    f32x4 bt[2][NT];
    SESSD_WG_LOAD(0, 0)
    for (int x0 = 0; x0 < A.wo; x0 += 16) {
      SESSD_WG_LOAD(1, x0 + 8)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_WG_MMA(0)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_WG_LOAD(0, x0 + 16)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_WG_MMA(1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SESSD_WG_LOAD
#undef SESSD_WG_MMA
  }
  // D layout: column (ci) = lane & 31, row (co) = (e & 3) + 8 * (e >> 2) + 4 * h
  float* dst = A.partial + (size_t)blockIdx.y * A.co * A.ci * NT;
  if (ci_ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int cor = cob * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (cor < A.co) dst[((size_t)cor * A.ci + ci) * NT + t] = acc[t][e];
      }
  }
}

// 1x1 convs with >= 64 channels on both sides (trans_0 / trans_1 of the SSFA neck): a plain GEMM dW[co][ci] = sum over pixels of
// g[co][p] x[ci][p]. The general kernel above has ONE accumulator per wave for a 1x1 (two 16-byte loads per four MFMAs:
// load-bound, 47 TFLOP/s); here a wave owns a 64 co x 64 ci block (2 x 2 accumulators: four loads per sixteen MFMAs), the row
// chunks are four times finer to keep the chip filled. Same lane layout and summation rule (row chunks summed in order).
__global__ __launch_bounds__(64) void conv1x1_wgrad_partial_kernel(WgArgs A) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int cib_n = sessd_divup(A.ci, 64);
  const int cob = blockIdx.x / cib_n, cib = blockIdx.x - cob * cib_n;
  const int R = A.batch * A.ho;
  const int r0 = blockIdx.y * A.rows_per_chunk, r1 = min(R, r0 + A.rows_per_chunk);
  const rsrc_t gr = make_rsrc(A.gout, (unsigned)((size_t)A.batch * A.co * A.ho * A.wo * 4));
  const rsrc_t xr = make_rsrc(A.inp, (unsigned)((size_t)A.batch * A.ci * A.hi * A.wi * 4));
  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  for (int r = r0; r < r1; ++r) {
    const int b = r / A.ho, y = r - b * A.ho;
    unsigned grow[2], xrow[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int co = cob * 64 + m * 32 + i, ci = cib * 64 + m * 32 + i;
      grow[m] = co < A.co ? (unsigned)((((size_t)b * A.co + co) * A.ho + y) * A.wo * 4) : SESSD_OOB;
      xrow[m] = ci < A.ci ? (unsigned)((((size_t)b * A.ci + ci) * A.hi + y) * A.wi * 4) : SESSD_OOB;
    }
    f32x4 a[2][2], bt[2][2];   // [set][block]
#define SESSD_W1_LOAD(SET, X0)                                                                     \
  {                                                                                                \
    const unsigned xo = (unsigned)((X0) + 4 * h) * 4u;                                             \
    const bool in = (X0) < A.wo;                                                                   \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                \
      a[SET][m] = ld4(gr, (grow[m] == SESSD_OOB || !in) ? SESSD_OOB : grow[m] + xo);               \
      bt[SET][m] = ld4(xr, (xrow[m] == SESSD_OOB || !in) ? SESSD_OOB : xrow[m] + xo);              \
    }                                                                                              \
  }
#define SESSD_W1_MMA(SET)                                                                          \
  _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                    \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[SET][m][c], bt[SET][n][c], acc[m][n], 0, 0, 0);
    SESSD_W1_LOAD(0, 0)
    for (int x0 = 0; x0 < A.wo; x0 += 16) {
      SESSD_W1_LOAD(1, x0 + 8)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_W1_MMA(0)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_W1_LOAD(0, x0 + 16)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_W1_MMA(1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SESSD_W1_LOAD
#undef SESSD_W1_MMA
  }
  float* dst = A.partial + (size_t)blockIdx.y * A.co * A.ci;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int ci = cib * 64 + n * 32 + i;
      if (ci < A.ci) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int cor = cob * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (cor < A.co) dst[(size_t)cor * A.ci + ci] = acc[m][n][e];
        }
      }
    }
}

// 3x3 stride-2 layers with cout, cin % 64 == 0 (b1.0 of the SSFA neck and, roles swapped, its two transposed convs), round 4:
// the kernel above gives every wave private operands -- a 32-ci input strip is fetched by each of the cout / 32 waves that
// need it, 7 x 16-byte + 3 x 4-byte loads per lane in front of every 36 MFMAs, two register sets deep: 71 TFLOP/s, the matrix
// cores 45 % busy (profiles/r4_kernel_trace_train_replay.txt: 293 us per launch). Here the operands of a step go through LDS
// ONCE per workgroup:
//   workgroup = 4 waves, a 64 co x 64 ci block (wave (m, n) owns co 32 m.., ci 32 n.. with all 9 taps: 9 accumulators)
//   stage     = 8 output pixels of one output row: A = gout[64 co][8 px] (128 x 16-byte loads), B = the three input rows
//               2y-1 .. 2y+1, 64 ci x 24 columns 2 x0 - 4 .. 2 x0 + 19 (1152 aligned 16-byte loads; rows / columns outside the
//               image by out-of-range buffer offsets = zero padding). A thread's <= 6 loads of stage s+1 are issued before the 36
//               MFMAs of stage s and written to the other LDS buffer after them: one barrier per stage.
//   MFMA step = pixel pair {2c, 2c+1}: a = A[co][2c + h], b_tap = B[ky][ci][2 (2c + h) + kx + 3] (odd row pitches: the 32 rows
//               of a wave hit 32 banks)
// Row chunks and their ordered sum as above (same partial layout, same reduce kernel). Workgroup id -> (chunk, block) keeps the
// blocks of one chunk on ONE XCD (id % 8 = chunk % 8): the chunk's rows are fetched into that XCD's L2 once.
constexpr int WL_PA = 9, WL_PB = 25;                       // LDS row pitches (floats)
constexpr int WL_A = 64 * WL_PA, WL_B = 3 * 64 * WL_PB;    // floats per buffer
__global__ __launch_bounds__(256, 2) void conv3x3s2_wgrad_lds_kernel(WgArgs A, int nchunks, int tiles) {
  __shared__ float sA[2][WL_A];
  __shared__ float sB[2][WL_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int tile = local % tiles, chunk = xcd + 8 * (local / tiles);
  if (chunk >= nchunks) return;
  const int cob = tile / A.cib_n, cib = tile - cob * A.cib_n;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int R = A.batch * A.ho;
  const int r0 = chunk * A.rows_per_chunk, r1 = min(R, r0 + A.rows_per_chunk);
  const int spr = A.wo >> 3;                               // stages per row
  const int S = (r1 - r0) * spr;
  const rsrc_t gr = make_rsrc(A.gout, (unsigned)((size_t)A.batch * A.co * A.ho * A.wo * 4));
  const rsrc_t xr = make_rsrc(A.inp, (unsigned)((size_t)A.batch * A.ci * A.hi * A.wi * 4));
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  // this thread's load slots: A (tid < 128): co = tid >> 1, 4-pixel half = tid & 1; B slot k: e = tid + 256 k < 1152 ->
  // row e / 384, ci (e % 384) / 6, 4-column chunk (e % 384) % 6
  const int a_co = tid >> 1, a_c = tid & 1;
  int b_row[5], b_ci[5], b_c[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int e = tid + 256 * k;
    b_row[k] = e / 384;
    const int rem = e - b_row[k] * 384;
    b_ci[k] = rem / 6;
    b_c[k] = rem - b_ci[k] * 6;
  }
  // Two register sets: the loads of stage s + 2 / s + 3 are in flight while stage s computes (one stage of 36 MFMAs is shorter
  // than a memory round trip; with a single set in flight the kernel ran at the private-operand kernel's 290 us). The loader
  // walks (batch, row, x0) incrementally; stages past the chunk load zeros.
  f32x4 ra[2], rb[2][5];
  int l_b = r0 / A.ho, l_y = r0 - l_b * A.ho, l_x = 0, l_left = S;
#define SESSD_WL_LOAD(SET)                                                                                          \
  {                                                                                                                 \
    const bool live_ = l_left > 0;                                                                                  \
    ra[SET] = ld4(gr, (live_ && tid < 128)                                                                          \
                          ? (unsigned)(((((size_t)l_b * A.co + co0 + a_co) * A.ho + l_y) * A.wo + l_x + 4 * a_c) * 4) \
                          : SESSD_OOB);                                                                             \
    _Pragma("unroll") for (int k = 0; k < 5; ++k) {                                                                 \
      const int yin = 2 * l_y + b_row[k] - 1, xin = 2 * l_x - 4 + 4 * b_c[k];                                       \
      const bool ok = live_ && (k < 4 || tid < 128) && yin >= 0 && yin < A.hi && xin >= 0 && xin < A.wi;            \
      rb[SET][k] = ld4(xr, ok ? (unsigned)(((((size_t)l_b * A.ci + ci0 + b_ci[k]) * A.hi + yin) * A.wi + xin) * 4) : SESSD_OOB); \
    }                                                                                                               \
    --l_left;                                                                                                       \
    l_x += 8;                                                                                                       \
    if (l_x == A.wo) {                                                                                              \
      l_x = 0;                                                                                                      \
      if (++l_y == A.ho) { l_y = 0; ++l_b; }                                                                        \
    }                                                                                                               \
  }
#define SESSD_WL_STORE(BUF, SET)                                                                                    \
  {                                                                                                                 \
    if (tid < 128) {                                                                                                \
      float* d = &sA[BUF][a_co * WL_PA + 4 * a_c];                                                                  \
      d[0] = ra[SET].x; d[1] = ra[SET].y; d[2] = ra[SET].z; d[3] = ra[SET].w;                                       \
    }                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < 5; ++k)                                                                   \
      if (k < 4 || tid < 128) {                                                                                     \
        float* d = &sB[BUF][(b_row[k] * 64 + b_ci[k]) * WL_PB + 4 * b_c[k]];                                        \
        d[0] = rb[SET][k].x; d[1] = rb[SET][k].y; d[2] = rb[SET][k].z; d[3] = rb[SET][k].w;                         \
      }                                                                                                             \
  }
#define SESSD_WL_MMA(BUF)                                                                                           \
  {                                                                                                                 \
    const float* pa = &sA[BUF][(wm * 32 + i) * WL_PA + h];                                                          \
    const float* pb = &sB[BUF][(wn * 32 + i) * WL_PB + 2 * h + 3];                                                  \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                 \
      const float a = pa[2 * c];                                                                                    \
      _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                              \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                          \
          const float bq = pb[ky * 64 * WL_PB + 4 * c + kx];                                                        \
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[ky * 3 + kx], 0, 0, 0);                \
        }                                                                                                           \
    }                                                                                                               \
  }
// LDS-only barrier: __syncthreads() is a workgroup-scope fence and hipcc puts s_waitcnt vmcnt(0) in front of it, i.e. it would
// wait for the loads that are meant to stay in flight across it
#define SESSD_WL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  SESSD_WL_LOAD(0)
  SESSD_WL_LOAD(1)
  SESSD_WL_STORE(0, 0)
  SESSD_WL_LOAD(0)
  SESSD_WL_BARRIER();
  for (int s = 0; s < S; s += 2) {
    SESSD_WL_MMA(0)
    __builtin_amdgcn_sched_barrier(0);
    SESSD_WL_STORE(1, 1)
    SESSD_WL_LOAD(1)
    SESSD_WL_BARRIER();
    if (s + 1 < S) SESSD_WL_MMA(1)
    __builtin_amdgcn_sched_barrier(0);
    SESSD_WL_STORE(0, 0)
    SESSD_WL_LOAD(0)
    SESSD_WL_BARRIER();
  }
#undef SESSD_WL_LOAD
#undef SESSD_WL_STORE
#undef SESSD_WL_MMA
#undef SESSD_WL_BARRIER
  // D layout: column (ci) = lane & 31, row (co) = (e & 3) + 8 * (e >> 2) + 4 * h
  float* dst = A.partial + (size_t)chunk * A.co * A.ci * 9;
  const int ci = ci0 + wn * 32 + i;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int cor = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      dst[((size_t)cor * A.ci + ci) * 9 + t] = acc[t][e];
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int nchunks, int total,
                                                                 float* __restrict__ gw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  // 16 loads in flight, added in chunk order (a rolled loop is one memory round trip per chunk: 17 us whatever the size)
  float s = 0.f;
  for (int c0 = 0; c0 < nchunks; c0 += 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = c0 + k < nchunks ? partial[(size_t)(c0 + k) * total + e] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
  }
  gw[e] = s;
}

constexpr int WG_MAX_CHUNKS = 64;
constexpr int W1_MAX_CHUNKS = 256;   // conv1x1_wgrad_partial_kernel: 64 x 64 wave tiles, finer row chunks

}  // namespace

extern "C" {

size_t sessd_conv2d_wgrad_workspace_bytes(int cout, int cin, int ksize) {
  return (size_t)(ksize == 1 ? W1_MAX_CHUNKS : WG_MAX_CHUNKS) * cout * cin * ksize * ksize * sizeof(float);
}

// grad_weight (cout, cin, k, k) of Conv2d(cin, cout, k, stride, padding = k/2, bias-free part): input (B, cin, hin, win),
// grad_out (B, cout, hout, wout) with hout = (hin + 2*(k/2) - k)/stride + 1. k in {1, 3}, stride in {1, 2} (k = 1: stride 1);
// wout % 8 == 0. For ConvTranspose2d(3, stride 2, padding 1, output_padding 1) call it with the roles swapped (input :=
// the transposed conv's grad_out, grad_out := its input): the result is that layer's (Cin, Cout, 3, 3) weight gradient.
int sessd_conv2d_wgrad(const float* input, int batch, int cin, int hin, int win, const float* grad_out, int cout, int hout,
                       int wout, int ksize, int stride, float* grad_weight, void* workspace, size_t workspace_bytes,
                       hipStream_t stream) {
  if (batch < 1 || cin < 1 || cout < 1 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return SESSD_EINVAL;
  if (ksize == 1 && stride != 1) return SESSD_EINVAL;
  const int p = ksize / 2;
  if (hout != (hin + 2 * p - ksize) / stride + 1 || wout != (win + 2 * p - ksize) / stride + 1) return SESSD_EINVAL;
  if (wout % 8 || (stride == 2 && win != 2 * wout)) return SESSD_EINVAL;
  if ((size_t)batch * cin * hin * win * 4 >= 0x7FFFFFFFull || (size_t)batch * cout * hout * wout * 4 >= 0x7FFFFFFFull)
    return SESSD_EINVAL;
  if (workspace_bytes < sessd_conv2d_wgrad_workspace_bytes(cout, cin, ksize)) return SESSD_EWORKSPACE;
  WgArgs A;
  A.inp = input; A.gout = grad_out; A.partial = (float*)workspace;
  A.batch = batch; A.ci = cin; A.hi = hin; A.wi = win; A.co = cout; A.ho = hout; A.wo = wout;
  if (ksize == 1 && cout >= 64 && cin >= 64) {
    const int tiles = sessd_divup(cout, 64) * sessd_divup(cin, 64), rows = batch * hout;
    int nchunks = 1024 / tiles;
    nchunks = nchunks < 8 ? 8 : (nchunks > W1_MAX_CHUNKS ? W1_MAX_CHUNKS : nchunks);
    if (nchunks > rows) nchunks = rows;
    A.rows_per_chunk = sessd_divup(rows, nchunks);
    nchunks = sessd_divup(rows, A.rows_per_chunk);
    A.cib_n = sessd_divup(cin, 64);
    SESSD_LAUNCH(conv1x1_wgrad_partial_kernel, dim3(tiles, nchunks), dim3(64), 0, stream, A);
    SESSD_CHECK_LAUNCH();
    const int total = cout * cin;
    SESSD_LAUNCH(conv_wgrad_reduce_kernel, dim3(sessd_divup(total, 256)), dim3(256), 0, stream, (const float*)workspace, nchunks,
                 total, grad_weight);
    SESSD_CHECK_LAUNCH();
    return SESSD_OK;
  }
  // SESSD_WGRAD_S2_LDS=0 keeps the private-operand kernel for A/B measurements
  static const bool s2_lds = !(getenv("SESSD_WGRAD_S2_LDS") && atoi(getenv("SESSD_WGRAD_S2_LDS")) == 0);
  if (s2_lds && ksize == 3 && stride == 2 && cout % 64 == 0 && cin % 64 == 0) {
    A.cib_n = cin / 64;
    const int tiles = (cout / 64) * A.cib_n, rows = batch * hout;
    int nchunks = 512 / tiles;   // two workgroups per CU
    nchunks = nchunks < 8 ? 8 : (nchunks > WG_MAX_CHUNKS ? WG_MAX_CHUNKS : nchunks);
    if (nchunks > rows) nchunks = rows;
    A.rows_per_chunk = sessd_divup(rows, nchunks);
    nchunks = sessd_divup(rows, A.rows_per_chunk);
    SESSD_LAUNCH(conv3x3s2_wgrad_lds_kernel, dim3(tiles * ((nchunks + 7) / 8 * 8)), dim3(256), 0, stream, A, nchunks, tiles);
    SESSD_CHECK_LAUNCH();
    const int total = cout * cin * 9;
    SESSD_LAUNCH(conv_wgrad_reduce_kernel, dim3(sessd_divup(total, 256)), dim3(256), 0, stream, (const float*)workspace, nchunks,
                 total, grad_weight);
    SESSD_CHECK_LAUNCH();
    return SESSD_OK;
  }
  const int cob_n = sessd_divup(cout, 32);
  A.cib_n = sessd_divup(cin, 32);
  const int tiles = cob_n * A.cib_n, rows = batch * hout;
  int nchunks = 2048 / tiles;
  nchunks = nchunks < 8 ? 8 : (nchunks > WG_MAX_CHUNKS ? WG_MAX_CHUNKS : nchunks);
  if (nchunks > rows) nchunks = rows;
  A.rows_per_chunk = sessd_divup(rows, nchunks);
  nchunks = sessd_divup(rows, A.rows_per_chunk);
  dim3 grid(tiles, nchunks);
  if (ksize == 1)
    SESSD_LAUNCH((conv_wgrad_partial_kernel<1, 1>), grid, dim3(64), 0, stream, A);
  else if (stride == 1)
    SESSD_LAUNCH((conv_wgrad_partial_kernel<3, 1>), grid, dim3(64), 0, stream, A);
  else
    SESSD_LAUNCH((conv_wgrad_partial_kernel<3, 2>), grid, dim3(64), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  const int total = cout * cin * ksize * ksize;
  SESSD_LAUNCH(conv_wgrad_reduce_kernel, dim3(sessd_divup(total, 256)), dim3(256), 0, stream, (const float*)workspace,
                     nchunks, total, grad_weight);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
