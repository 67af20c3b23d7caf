// ConvTranspose2d(cin, cout, 3, stride 2, padding 1, output_padding 1) + folded BatchNorm + ReLU (+ residual) of the SSFA neck
// (det3d/models/necks/rpn_v1.py:175-199, 224-226), tile_cfg 40 .. 42 of sessd_deconv2d_s2_mfma: the two output-parity classes
// px = 0 / 1 of an output row parity py are computed TOGETHER by every wave, into two accumulators.
//
// The class-per-workgroup launch (conv2d_mfma4_kernel) writes every output line twice, 4 bytes out of every 8 each time (the
// classes px = 0 and px = 1 interleave along x), and loads the same input pixels once per class. Here lane j of a wave owns the
// tile-space pixel (y, x) for BOTH px: out(2y+py, 2x) and out(2y+py, 2x+1) leave as one 8-byte store (32 lanes = 256 contiguous
// bytes = whole lines, written once), the residual is read the same way, and the input values in(y+ey, x), in(y+ey, x+1) are
// loaded once and multiplied by the weights of both classes:
//     py = 0: acc0 += W[1][1] in(y,x)                        acc1 += W[1][0] in(y,x+1) + W[1][2] in(y,x)
//     py = 1: acc0 += W[0][1] in(y+1,x) + W[2][1] in(y,x)    acc1 += W[0][0] in(y+1,x+1) + W[0][2] in(y+1,x) + W[2][0] in(y,x+1) + W[2][2] in(y,x)
// -> 5 operand loads per 3 MFMAs instead of 6 (the direct kernels are bound by the L1 fill rate, not by the matrix cores).
// Same direct-to-register scheme as conv_body (dense_conv.hip): buffer loads with SGPR offsets, no VALU in the k-loop, two
// operand register sets. The weights are the four per-class packings of ops.pack_deconv2d_s2 ([cin/2][ntaps][2][cout_pad]).
// Numerics: per class the same fmaf chain over (cin pair, tap, cin parity) as conv2d_mfma4_kernel -- bit-identical results.
#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2v = __attribute__((ext_vector_type(2))) float;
typedef unsigned int u32x2g __attribute__((__vector_size__(8)));

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bufload(rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}
#define SESSD_OOB 0x80000000u

struct DeconvPairArgs {
  const float* in;        // (B, cin, hin, win)
  float* out;             // (B, cout, 2 hin, 2 win)
  const float* scale;
  const float* shift;
  const float* residual;
  const float* w[4];      // classes (py,px) = (0,0),(0,1),(1,0),(1,1): [cin/2][1|2|2|4][2][cout_pad]
  int cin, hin, win, cout, cout_pad, relu;
};

// PY = output row parity. NX = input pixels per lane and cin pair (2 or 4), NT0 / NT1 = taps of the px = 0 / px = 1 class,
// G = cin pairs per k-step (so that every step issues 10 loads and 6 MFMAs).
template <int PY, int NW, int CT>
__device__ __forceinline__ void deconv_pair_body(const DeconvPairArgs& A, const int b) {
  constexpr int NT0 = PY ? 2 : 1, NT1 = PY ? 4 : 2, NX = PY ? 4 : 2, G = PY ? 1 : 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int npix = A.hin * A.win;
  const int ny = sessd_divup(A.cout_pad, CT * 32);  // cout groups of CT x 32
  // XCD-aware workgroup order (see conv_body): each XCD owns a contiguous run of pixel tiles with all their cout groups
  int bx, by;
  {
    const int total = gridDim.x, bid = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = bid & 7, loc = bid >> 3;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    bx = wgid / ny;
    by = wgid - bx * ny;
  }
  const int p_base = (bx * NW + wave) * 32, m_base = by * (CT * 32);
  if (p_base >= npix) return;
  const int in_plane = npix;
  const int p = p_base + j;
  const bool live = p < npix;
  const int y = live ? p / A.win : 0, x = live ? p - (p / A.win) * A.win : 0;
  // input pixels (y + ey, x + ex): index e = ey * 2 + ex
  unsigned xo[NX];
#pragma unroll
  for (int e = 0; e < NX; ++e) {
    const int iy = y + (e >> 1), ix = x + (e & 1);
    xo[e] = (live && iy < A.hin && ix < A.win) ? (unsigned)((h * in_plane + iy * A.win + ix) * 4) : SESSD_OOB;
  }
  const unsigned plane8 = 2u * (unsigned)in_plane * 4u;  // bytes per cin pair
  const rsrc_t xr = make_rsrc(A.in + (size_t)b * A.cin * in_plane, (unsigned)A.cin * in_plane * 4u);
  const unsigned wrow = 2u * (unsigned)A.cout_pad * 4u;  // weight bytes per (cin pair, tap)
  const rsrc_t w0r = make_rsrc(A.w[PY * 2], (unsigned)(A.cin >> 1) * NT0 * wrow);
  const rsrc_t w1r = make_rsrc(A.w[PY * 2 + 1], (unsigned)(A.cin >> 1) * NT1 * wrow);
  const unsigned wo = (unsigned)((h * A.cout_pad + m_base + j) * 4);

  f32x16 acc0[CT], acc1[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[c][r] = 0.f; acc1[c][r] = 0.f; }
  const int KS = (A.cin >> 1) / G;  // k-steps
  float xv[2][G][NX], w0v[2][G][NT0][CT], w1v[2][G][NT1][CT];

#define SESSD_DP_LOAD(SET, STEP)                                                                    \
  {                                                                                                 \
    _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                 \
      const unsigned kp = (unsigned)((STEP)*G + g);                                                 \
      _Pragma("unroll") for (int e = 0; e < NX; ++e) xv[SET][g][e] = bufload(xr, xo[e], kp * plane8); \
      _Pragma("unroll") for (int t = 0; t < NT0; ++t)                                               \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) w0v[SET][g][t][c] = bufload(w0r, wo + c * 128u, (kp * NT0 + t) * wrow); \
      _Pragma("unroll") for (int t = 0; t < NT1; ++t)                                               \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) w1v[SET][g][t][c] = bufload(w1r, wo + c * 128u, (kp * NT1 + t) * wrow); \
    }                                                                                               \
  }
  // tap t of the px = 0 class reads pixel index X0[t], of the px = 1 class X1[t] (class tap order of ops.pack_deconv2d_s2)
#define SESSD_DP_M(ACC, WV, T, E) \
  _Pragma("unroll") for (int c = 0; c < CT; ++c) ACC[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(WV[SET_][g][T][c], xv[SET_][g][E], ACC[c], 0, 0, 0);
#define SESSD_DP_MMA(SET)                                                                           \
  {                                                                                                 \
    constexpr int SET_ = SET;                                                                       \
    _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                 \
      if constexpr (PY == 0) {                                                                      \
        SESSD_DP_M(acc0, w0v, 0, 0)                                                                 \
        SESSD_DP_M(acc1, w1v, 0, 1)                                                                 \
        SESSD_DP_M(acc1, w1v, 1, 0)                                                                 \
      } else {                                                                                      \
        SESSD_DP_M(acc0, w0v, 0, 2)                                                                 \
        SESSD_DP_M(acc1, w1v, 0, 3)                                                                 \
        SESSD_DP_M(acc0, w0v, 1, 0)                                                                 \
        SESSD_DP_M(acc1, w1v, 1, 2)                                                                 \
        SESSD_DP_M(acc1, w1v, 2, 1)                                                                 \
        SESSD_DP_M(acc1, w1v, 3, 0)                                                                 \
      }                                                                                             \
    }                                                                                               \
  }
  // every load unconditional (the last one clamped and unused), "issue next set, then consume the current one" pinned
  SESSD_DP_LOAD(0, 0)
  for (int ks = 0; ks + 2 <= KS; ks += 2) {
    SESSD_DP_LOAD(1, ks + 1)
    __builtin_amdgcn_sched_barrier(0);
    SESSD_DP_MMA(0)
    __builtin_amdgcn_sched_barrier(0);
    SESSD_DP_LOAD(0, min(ks + 2, KS - 1))
    __builtin_amdgcn_sched_barrier(0);
    SESSD_DP_MMA(1)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (KS & 1) SESSD_DP_MMA(0)
#undef SESSD_DP_LOAD
#undef SESSD_DP_MMA
#undef SESSD_DP_M

  // epilogue, branch-free: scale / shift / residual loads of all 16 rows in flight together, 8-byte stores; an element
  // outside the image or beyond cout gets an out-of-range buffer offset. D layout: column = lane & 31 (pixel),
  // row = (r & 3) + 8 (r >> 2) + 4 h (cout).
  const unsigned oplane = 4u * (unsigned)npix;  // output plane in floats: (2 hin) x (2 win)
  const size_t boff = (size_t)b * A.cout * oplane;
  const rsrc_t orr = make_rsrc(A.out + boff, (unsigned)A.cout * oplane * 4u);
  const rsrc_t rr = make_rsrc(A.residual ? A.residual + boff : A.out, A.residual ? (unsigned)A.cout * oplane * 4u : 0u);
  const rsrc_t scr = make_rsrc(A.scale ? A.scale : A.out, A.scale ? (unsigned)A.cout * 4u : 0u);
  const rsrc_t shr = make_rsrc(A.shift ? A.shift : A.out, A.shift ? (unsigned)A.cout * 4u : 0u);
  const unsigned pix4 = (unsigned)((2 * y + PY) * (2 * A.win) + 2 * x) * 4u;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int co0 = m_base + c * 32 + 4 * h;
    const unsigned vbase = (unsigned)co0 * oplane * 4u + pix4;
    float scv[16], shv[16];
    unsigned vo[16];
    f32x2v rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2);
      scv[r] = bufload(scr, (unsigned)(co0 + k) * 4u, 0);
      shv[r] = bufload(shr, (unsigned)(co0 + k) * 4u, 0);
      vo[r] = (live && co0 + k < A.cout) ? vbase + (unsigned)k * oplane * 4u : SESSD_OOB;
      rv[r] = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(rr, (int)vo[r], 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sc = A.scale ? scv[r] : 1.f;
      float v0 = fmaf(acc0[c][r], sc, shv[r]), v1 = fmaf(acc1[c][r], sc, shv[r]);
      if (A.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      f32x2v v;
      v.x = A.residual ? v0 + rv[r].x : v0;
      v.y = A.residual ? v1 + rv[r].y : v1;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2g, v), orr, (int)vo[r], 0, 0);
    }
  }
}

// blockIdx.z = batch * 2 + (1 - py): the py = 1 half (twice the work per workgroup) is dispatched first
template <int NW, int CT>
__global__ __launch_bounds__(NW * 64) void deconv_s2_pair_kernel(DeconvPairArgs A) {
  if ((blockIdx.z & 1) == 0)
    deconv_pair_body<1, NW, CT>(A, blockIdx.z >> 1);
  else
    deconv_pair_body<0, NW, CT>(A, blockIdx.z >> 1);
}

}  // namespace

// variant = workgroup shape: 0: 4 waves (128 pixels) x 32 couts, 1: 2 waves x 32 couts, 2: 1 wave x 32 couts (finer shapes balance
// the launch better: the py = 1 workgroups carry twice the work of the py = 0 ones). A 64-cout wave (CT = 2: every input value
// feeds two cout tiles, 4 operand loads per 3 MFMAs) was measured slower (236 registers, half the waves in flight).
// Called by sessd_deconv2d_s2_mfma (dense_conv.hip) for tile_cfg 40 + variant; wpk4 in class order (0,0),(0,1),(1,0),(1,1).
int sessd_deconv_pair_launch(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4, float* out, int cout,
                             const float* scale, const float* shift, int relu, const float* residual, int variant,
                             hipStream_t stream) {
  if (cin % 4 || batch < 1 || cout < 1 || variant < 0 || variant > 2) return SESSD_EINVAL;
  if ((size_t)cout * 4 * hin * win * 4 > 0x7fffffffULL || (size_t)cin * hin * win * 4 > 0x7fffffffULL) return SESSD_EINVAL;
  DeconvPairArgs A;
  A.in = in; A.out = out; A.scale = scale; A.shift = shift; A.residual = residual;
  for (int c = 0; c < 4; ++c) A.w[c] = wpk4[c];
  A.cin = cin; A.hin = hin; A.win = win; A.cout = cout; A.cout_pad = sessd_divup(cout, 32) * 32; A.relu = relu;
  static const int nw[3] = {4, 2, 1};
  dim3 grid(sessd_divup(hin * win, nw[variant] * 32) * (A.cout_pad / 32), 1, batch * 2);
  switch (variant) {
    case 0: SESSD_LAUNCH((deconv_s2_pair_kernel<4, 1>), grid, dim3(256), 0, stream, A); break;
    case 1: SESSD_LAUNCH((deconv_s2_pair_kernel<2, 1>), grid, dim3(128), 0, stream, A); break;
    default: SESSD_LAUNCH((deconv_s2_pair_kernel<1, 1>), grid, dim3(64), 0, stream, A); break;
  }
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}
