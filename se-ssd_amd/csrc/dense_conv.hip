// Dense 2-D BEV convolutions of the SSFA neck and the MultiGroupHead 1x1 heads on gfx950.
// Replaces the cuDNN/MIOpen conv2d / conv_transpose2d + BatchNorm2d + ReLU chain of
//   det3d/models/necks/rpn_v1.py:135-210,220-235 (SSFA) and the four 1x1 convs of
//   det3d/models/bbox_heads/mg_head_sessd.py:202-230 (Head.forward).
//
// Direct-to-register implicit GEMM on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32):
//     D[cout][pixel] += W[cout][k] * X[k][pixel],   k = (cin pair, tap)
// A operand (weights): lane (i, h) = (lane&31, lane>>5) holds Wp[kp][tap][h][m0+i]  -- the packed
//     weight layout makes this two 128-byte rows per wave, L2-resident (<= 2.4 MB per layer).
// B operand (activations, NCHW): lane (j, h) holds X[2*kp+h][pixel p0+j shifted by the tap] -- 32
//     CONSECUTIVE pixels of one channel plane = one 128-byte line per half wave. Because the f32
//     MFMA issues only every 64 cycles per SIMD, operands go global/L2/L1 -> VGPR directly:
//     no LDS staging, no barriers; the 9 taps of a 3x3 window re-hit the same lines in L1.
// D (32 couts x 32 pixels per MFMA tile): lane holds pixel j and 16 couts -> every store is two
//     full 128-byte lines; epilogue fuses the folded BatchNorm (scale, shift), ReLU and an optional
//     residual add (deconv_block_0(x) + x_trans_0, rpn_v1.py:225).
// One kernel covers conv 3x3 s1, 3x3 s2, 1x1 and the 4 output-parity classes of the 3x3 s2
// transposed conv through (taps, in_mul, out_mul, out_py, out_px):
//     input pixel = (y*in_mul + dy[t], x*in_mul + dx[t]),  output pixel = (y*out_mul + py, x*out_mul + px)
// Numerics: bit-for-bit an fmaf chain over (cin pair, tap, cin parity) -- exact float32.
#include "common.hpp"
#include "sessd_hip_types.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct ConvArgs {
  const float* in;     // (B, cin, hin, win)
  const float* wpk;    // [cin/2][ntaps][2][cout_pad]
  float* out;          // (B, cout, hout, wout)
  const float* scale;  // [cout] or null
  const float* shift;  // [cout] or null
  const float* residual;  // same shape as out, or null (added after the activation)
  int cin, hin, win;
  int cout, cout_pad, hout, wout;
  int ht, wt;          // tile-space extent
  int in_mul, out_mul, out_py, out_px;
  int relu;
  int cgroup;          // cin pairs consumed per k-step (1 for 3x3; >1 batches 1x1 / few-tap convs into "virtual taps")
  int dy[9], dx[9];
  int dc[9];           // cin-pair displacement of a tap inside its k-step group
  // active-tile mode (LIST kernels; csrc/dense_active.hip): the tile-space pixels are the 2x2 tiles of a device list, entries
  // image * ntiles2 + tile, their count on the device; lbatch = images the buffers hold (the image is part of the lane offsets)
  const int* tile_list;
  const int* n_list;
  int list_cap, ntiles2, tw2, lbatch;
};

struct ConvArgs4 {
  ConvArgs c[4];
};
struct ConvArgs8 {
  ConvArgs c[8];
};

// wave tile: (CT*32 couts) x (PT*32 pixels); workgroup = 4 waves arranged WC x WP
// Raw buffer resource (V#) over [base, base + bytes): loads whose (voffset + imm) >= bytes return 0 in hardware.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bufload(rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}
#define SESSD_OOB 0x80000000u  // a lane offset beyond any buffer: the load returns 0 (out-of-image tap)

// LIST: slot p of the launch's pixel axis -> (image, y, x) through the tile list. 128 consecutive slots = 32 list entries: slots
// 0..63 the upper pixel rows of the 32 tiles, 64..127 the lower ones, so that a run of adjacent tiles is a run of adjacent pixels.
__device__ __forceinline__ bool list_pixel(const ConvArgs& A, int n_list, int p, int& y, int& x, int& img) {
  const int k = (p >> 7) * 32 + ((p & 63) >> 1);
  const int e = k < n_list ? A.tile_list[k] : -1;
  const bool live = e >= 0;
  img = live ? e / A.ntiles2 : 0;
  const int t = live ? e - img * A.ntiles2 : 0;
  y = 2 * (t / A.tw2) + ((p >> 6) & 1);
  x = 2 * (t - (t / A.tw2) * A.tw2) + (p & 1);
  return live;
}

template <int NTAPS, int CT, int PT, int WC, int WP, bool DEEP = false, bool LIST = false>
__device__ __forceinline__ void conv_body(const ConvArgs& A, const int b) {
  static_assert(WC * WP == 4, "four waves per workgroup");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wc = wave % WC, wp = wave / WC;
  int npix = A.ht * A.wt;
  int n_list = 0;
  if constexpr (LIST) {
    // ACTIVE-TILE mode: the launch is sized for the whole map, the workgroups beyond the device count leave at once
    n_list = __builtin_amdgcn_readfirstlane(min(A.n_list[0], A.list_cap));
    npix = ((n_list + 31) >> 5) * 128;
  }
  // XCD-aware workgroup order: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs (private
  // L2 each). Remap so that each XCD owns one contiguous run of pixel tiles (with all their cout groups): the
  // 3-row halo that neighbouring tiles share then hits in that XCD's L2 instead of being fetched once per XCD
  // (rocprofv3 FETCH_SIZE of the 128->128 layer: 131 MB -> see profiles/; the algorithmic input is 18.6 MB).
  const int ny = sessd_divup(A.cout_pad, WC * CT * 32);
  int bx, by;
  {
    const int total = LIST ? sessd_divup(npix, WP * PT * 32) * ny : (int)gridDim.x, bid = blockIdx.x;
    if (LIST && bid >= total) return;
    const int q = total >> 3, r = total & 7, xcd = bid & 7, loc = bid >> 3;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    bx = wgid / ny;
    by = wgid - bx * ny;
  }
  const int p_base = (bx * WP + wp) * (PT * 32);
  const int m_base = (by * WC + wc) * (CT * 32);
  if (p_base >= npix || m_base >= A.cout_pad) return;
  const int in_plane = A.hin * A.win;

  // Operands are fetched with BUFFER loads: address = SGPR resource + per-lane 32-bit offset (loop invariant)
  // + SGPR offset (advanced per k-step by the scalar unit) + immediate. The k-loop therefore issues NO vector
  // ALU work for addressing -- on gfx950 the f32 MFMA shares the SIMD's f32 lanes with the VALU, so every VALU
  // instruction in the loop is MFMA time lost (a 64-bit pointer add per load halved the rate of this kernel).
  // Out-of-image taps get the offset SESSD_OOB: the hardware range check returns 0, no select needed.
  const unsigned xbytes = (unsigned)A.cin * in_plane * 4u;
  const rsrc_t xr = LIST ? make_rsrc(A.in, (unsigned)A.lbatch * xbytes) : make_rsrc(A.in + (size_t)b * A.cin * in_plane, xbytes);
  const rsrc_t wr = make_rsrc(A.wpk, (unsigned)(A.cin >> 1) * NTAPS / A.cgroup * 2u * A.cout_pad * 4u);
  unsigned xo[PT][NTAPS];
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int p = p_base + q * 32 + j;
    bool live = p < npix;
    int y = live ? p / A.wt : 0, x = live ? p - (p / A.wt) * A.wt : 0, img = 0;
    if constexpr (LIST) live = list_pixel(A, n_list, p, y, x, img);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const int iy = y * A.in_mul + A.dy[t], ix = x * A.in_mul + A.dx[t];
      const bool ok = live && iy >= 0 && iy < A.hin && ix >= 0 && ix < A.win;
      xo[q][t] = ok ? (unsigned)img * xbytes + (unsigned)(((h + A.dc[t] * 2) * in_plane + iy * A.win + ix) * 4) : SESSD_OOB;
    }
  }
  const unsigned wo = (unsigned)((h * A.cout_pad + m_base + j) * 4);

  f32x16 acc[CT][PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int q = 0; q < PT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;

  const int KP = (A.cin >> 1) / A.cgroup;  // k-steps; the packed weight layout is the same for any cgroup
  const unsigned wstep = (unsigned)NTAPS * 2u * A.cout_pad * 4u;       // weight bytes per k-step
  const unsigned wtap = 2u * A.cout_pad * 4u;                          // weight bytes per tap
  const unsigned xstep = 2u * (unsigned)A.cgroup * (unsigned)in_plane * 4u;  // activation bytes per k-step

  float wa[DEEP ? 3 : 2][NTAPS][CT], xb[DEEP ? 3 : 2][NTAPS][PT];

#define SESSD_LOAD(SET, KPI)                                                                   \
  {                                                                                            \
    const unsigned ws = (unsigned)(KPI)*wstep, xs = (unsigned)(KPI)*xstep;                      \
    _Pragma("unroll") for (int t = 0; t < NTAPS; ++t) {                                        \
      _Pragma("unroll") for (int c = 0; c < CT; ++c) wa[SET][t][c] = bufload(wr, wo + c * 128u, ws + t * wtap); \
      _Pragma("unroll") for (int q = 0; q < PT; ++q) xb[SET][t][q] = bufload(xr, xo[q][t], xs); \
    }                                                                                          \
  }
#define SESSD_MMA(SET)                                                                         \
  {                                                                                            \
    _Pragma("unroll") for (int t = 0; t < NTAPS; ++t)                                          \
      _Pragma("unroll") for (int q = 0; q < PT; ++q)                                           \
        _Pragma("unroll") for (int c = 0; c < CT; ++c)                                         \
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[SET][t][c], xb[SET][t][q], acc[c][q], 0, 0, 0); \
  }

  // Two register sets, one cin pair of look-ahead. Every load in the loop is UNCONDITIONAL (the last one is
  // clamped and unused): a load under a branch makes hipcc's vmcnt bookkeeping conservative at the merge and it
  // then waits for the loads it has just issued -- the whole memory latency, every iteration.
  if constexpr (!DEEP) {
    SESSD_LOAD(0, 0)
    for (int kp = 0; kp + 2 <= KP; kp += 2) {
      SESSD_LOAD(1, kp + 1)
      __builtin_amdgcn_sched_barrier(0);  // keep "issue next set, then consume current set" in program order
      SESSD_MMA(0)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_LOAD(0, min(kp + 2, KP - 1))
      __builtin_amdgcn_sched_barrier(0);
      SESSD_MMA(1)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KP & 1) SESSD_MMA(0)  // odd KP tail (set 0 holds KP-1)
  } else {
    // three register sets = two k-steps of look-ahead: a wave hides an L2 round trip by itself, which matters in the
    // tail of a batch-1 launch when few waves are left on a SIMD
    SESSD_LOAD(0, 0)
    SESSD_LOAD(1, min(1, KP - 1))
    int kp = 0;
    for (; kp + 3 <= KP; kp += 3) {
      SESSD_LOAD(2, min(kp + 2, KP - 1))
      __builtin_amdgcn_sched_barrier(0);
      SESSD_MMA(0)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_LOAD(0, min(kp + 3, KP - 1))
      __builtin_amdgcn_sched_barrier(0);
      SESSD_MMA(1)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_LOAD(1, min(kp + 4, KP - 1))
      __builtin_amdgcn_sched_barrier(0);
      SESSD_MMA(2)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kp < KP) SESSD_MMA(0)      // set 0 holds kp
    if (kp + 1 < KP) SESSD_MMA(1)  // set 1 holds kp+1
  }
#undef SESSD_LOAD
#undef SESSD_MMA

  // epilogue. D layout (32x32): column = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*h (cout).
  // Branch-free: scale / shift / residual / output go through buffer resources and an element outside the image or beyond cout
  // gets an out-of-range offset (loads return 0, stores are dropped), so that the 48 loads of a 32x32 tile are in flight together.
  // (With `if (co < cout)` / `if (residual)` around each element hipcc emitted load, s_waitcnt vmcnt(0), load, s_waitcnt vmcnt(0),
  // store per element: 32 serialised memory round trips per tile, each also waiting for the previous store.)
  const unsigned oplane4 = (unsigned)(A.hout * A.wout) * 4u;
  const size_t boff = LIST ? (size_t)0 : (size_t)b * A.cout * (size_t)(A.hout * A.wout);
  const unsigned obytes1 = (unsigned)A.cout * oplane4;   // one image's output
  const unsigned obytes = LIST ? (unsigned)A.lbatch * obytes1 : obytes1;
  const rsrc_t orr = make_rsrc(A.out + boff, obytes);
  const rsrc_t rr = make_rsrc(A.residual ? A.residual + boff : A.out, A.residual ? obytes : 0u);
  const rsrc_t scr = make_rsrc(A.scale ? A.scale : A.out, A.scale ? (unsigned)A.cout * 4u : 0u);
  const rsrc_t shr = make_rsrc(A.shift ? A.shift : A.out, A.shift ? (unsigned)A.cout * 4u : 0u);
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int p = p_base + q * 32 + j;
    bool live = p < npix;
    int y = live ? p / A.wt : 0, x = live ? p - (p / A.wt) * A.wt : 0, img = 0;
    if constexpr (LIST) live = list_pixel(A, n_list, p, y, x, img);
    const unsigned pix4 = (unsigned)img * obytes1 + (unsigned)((y * A.out_mul + A.out_py) * A.wout + (x * A.out_mul + A.out_px)) * 4u;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int co0 = m_base + c * 32 + 4 * h;
      const unsigned vbase = (unsigned)co0 * oplane4 + pix4;
      float scv[16], shv[16], rv[16];
      unsigned vo[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2);
        scv[r] = bufload(scr, (unsigned)(co0 + k) * 4u, 0);
        shv[r] = bufload(shr, (unsigned)(co0 + k) * 4u, 0);
        vo[r] = (live && co0 + k < A.cout) ? vbase + (unsigned)k * oplane4 : SESSD_OOB;
        rv[r] = bufload(rr, vo[r], 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = fmaf(acc[c][q][r], A.scale ? scv[r] : 1.f, shv[r]);
        if (A.relu) v = fmaxf(v, 0.f);
        if (A.residual) v += rv[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, (int)vo[r], 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, ACTIVATION-STATIONARY variant: a workgroup = 32 pixels x 128 couts (4 waves x 32 couts).
// The nine shifted 32-pixel rows of 16 input channels (one "chunk" = 8 k-steps) are staged ONCE per workgroup in
// LDS as ready-made MFMA B operands [channel][tap][pixel] (zero padding already applied by the OOB buffer loads),
// so the per-wave global traffic is only the weight rows (aligned 128-B lines); the unaligned, 9x redundant
// activation loads of the direct kernel disappear (2 -> 1.25 VMEM instructions per MFMA, none unaligned).
// Double-buffered LDS, one barrier per chunk; the 18 staging loads of chunk c+1 are spread over the 8 k-steps of
// chunk c right behind that k-step's weight loads, so under the in-order vmcnt they get a full k-step of latency
// cover like every other load. Same (cin pair, tap, parity) fmaf order as conv_body: bit-identical results.
__global__ __launch_bounds__(256) void conv3x3s1_lds_kernel(ConvArgs A) {
  constexpr int KC = 8;            // k-steps (cin pairs) per chunk
  constexpr int CHF = 2 * KC * 288;  // floats per LDS buffer: 16 channels x 9 taps x 32 pixels
  __shared__ float xs[2][CHF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int npix = A.ht * A.wt;
  const int p_base = blockIdx.x * 32;
  const int m_base = (blockIdx.y * 4 + wave) * 32;
  const int b = blockIdx.z;
  const int in_plane = A.hin * A.win;
  const rsrc_t xr = make_rsrc(A.in + (size_t)b * A.cin * in_plane, (unsigned)A.cin * in_plane * 4u);
  const rsrc_t wr = make_rsrc(A.wpk, (unsigned)(A.cin >> 1) * 9u * 2u * A.cout_pad * 4u);
  // staging loads: flat element e = tid + 256*n of the chunk image [16 ch][9 taps][32 px]
  unsigned lo[18];
#pragma unroll
  for (int n = 0; n < 18; ++n) {
    const int e = tid + 256 * n;
    const int c = e / 288, pr = e - c * 288, t = pr >> 5, jj = pr & 31;
    const int p = p_base + jj;
    const bool live = p < npix;
    const int y = live ? p / A.wt : 0, x = live ? p - (p / A.wt) * A.wt : 0;
    const int iy = y + A.dy[t], ix = x + A.dx[t];
    const bool ok = live && iy >= 0 && iy < A.hin && ix >= 0 && ix < A.win;
    lo[n] = ok ? (unsigned)((c * in_plane + iy * A.win + ix) * 4) : SESSD_OOB;
  }
  const bool wave_live = m_base < A.cout_pad;  // waves beyond cout still stage and hit the barriers
  const unsigned wo = (unsigned)((h * A.cout_pad + (wave_live ? m_base : 0) + j) * 4);
  const unsigned wstep = 9u * 2u * A.cout_pad * 4u, wtap = 2u * A.cout_pad * 4u;
  const unsigned xchunk = 2u * KC * (unsigned)in_plane * 4u;
  const int KP = A.cin >> 1, NCH = KP / KC;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float wa[2][9], xb[2][9], xg[18];

#define SESSD_LOADW(SET, KPI)                                                                   \
  {                                                                                             \
    const unsigned ws = (unsigned)(KPI)*wstep;                                                  \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) wa[SET][t] = bufload(wr, wo, ws + t * wtap);  \
  }
#define SESSD_READB(SET, BUF, KPL)                                                              \
  {                                                                                             \
    const float* src = &xs[BUF][(2 * (KPL) + h) * 288 + j];                                     \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) xb[SET][t] = src[t * 32];                     \
  }
  // prologue: stage chunk 0, first weight set
#pragma unroll
  for (int n = 0; n < 18; ++n) xg[n] = bufload(xr, lo[n], 0);
  SESSD_LOADW(0, 0)
#pragma unroll
  for (int n = 0; n < 18; ++n) xs[0][tid + 256 * n] = xg[n];
  __syncthreads();

  for (int ch = 0; ch < NCH; ++ch) {
    const int cur = ch & 1;
    const unsigned xs_next = (unsigned)min(ch + 1, NCH - 1) * xchunk;
    SESSD_READB(0, cur, 0)
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int kp = ch * KC + s;
      SESSD_LOADW((s + 1) & 1, min(kp + 1, KP - 1))
      // this k-step's share of the next chunk's staging loads (18 over 8 steps: 2,2,2,2,2,2,3,3)
      {
        constexpr int lo_n[9] = {0, 2, 4, 6, 8, 10, 12, 15, 18};
#pragma unroll
        for (int n = lo_n[s]; n < lo_n[s + 1]; ++n) xg[n] = bufload(xr, lo[n], xs_next);
      }
      if (s + 1 < KC) SESSD_READB((s + 1) & 1, cur, s + 1)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s & 1][t], xb[s & 1][t], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 18; ++n) xs[cur ^ 1][tid + 256 * n] = xg[n];
    __syncthreads();
  }
#undef SESSD_LOADW
#undef SESSD_READB
  if (!wave_live) return;
  // epilogue (same as conv_body with CT = PT = 1)
  const size_t out_plane = (size_t)A.hout * A.wout;
  float* outb = A.out + (size_t)b * A.cout * out_plane;
  const float* resb = A.residual ? A.residual + (size_t)b * A.cout * out_plane : nullptr;
  const int p = p_base + j;
  if (p >= npix) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = m_base + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (co >= A.cout) continue;
    float v = acc[r];
    const float sc = A.scale ? A.scale[co] : 1.f, sh = A.shift ? A.shift[co] : 0.f;
    v = fmaf(v, sc, sh);
    if (A.relu) v = fmaxf(v, 0.f);
    if (resb) v += resb[(size_t)co * out_plane + p];
    outb[(size_t)co * out_plane + p] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution by fused Winograd F(2x2,3x3) on the f32 matrix cores (tile_cfg 20).
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A      d = 4x4 input patch, Y = 2x2 outputs, 16 products instead of 36
// i.e. 16 independent GEMMs  M_xi[cout][tile] = sum_cin U_xi[cout][cin] V_xi[cin][tile]  -> 2.25x fewer MFMAs than the
// direct kernel for the same layer. Everything is fused in one launch:
//   workgroup = 32 tiles (128 output pixels) x 32 couts, 4 waves; wave w owns the transform points xi = 4w..4w+3
//   round = 4 k-steps (8 input channels). In a round every wave (a) issues the patch loads (4 x 16 B per lane: lane =
//   (tile, channel parity)) and the U loads of the NEXT round, (b) runs its 16 MFMAs of THIS round with B operands
//   read from LDS (V of this round) and A operands (U) from registers, (c) transforms its patches (32 adds, in
//   registers) and writes V of the next round to the other LDS buffer; one barrier per round.
//   epilogue: the 16 M_xi of a (cout, tile) live in 4 waves -> through LDS, then the 2x2 output transform, BatchNorm,
//   ReLU, residual and 8-byte stores (consecutive lanes = consecutive tiles of one output row).
// U = G g G^T is precomputed on the host and packed [cin/2][wave 4][h 2][cout_pad][xi_local 4] (xi = 4*wave + xi_local):
// the four A operands of a lane and k-step are one 16-byte load.
// Zero padding: rows outside the image get an out-of-range buffer offset (hardware returns 0); the column left of
// x = 0 / right of x = W-1 is masked in registers. Needs even H and W and cin % 8 == 0.
// Numerics: NOT the fmaf chain of the direct kernel -- Winograd rounding (about 1e-6 of the output scale in float32);
// deterministic (fixed order), covered by the same tolerance as the other float stages.
// DEEP = false: patches and U are fetched one round ahead (2 U register sets, 1 patch set);
// DEEP = true : two rounds ahead (4 U sets, 2 patch sets; the round body is unrolled 4x so every set index is a
//               compile-time constant; cin % 32 == 0) -- the loads come from L2 (each workgroup streams 256 KB of patches and 256 KB
//               of U with no reuse), whose loaded latency exceeds one round of MFMA time.
template <bool DEEP>
__global__ __launch_bounds__(256, 2) void conv3x3s1_winograd_kernel(ConvArgs A) {
  constexpr int TT = 32;    // tiles per workgroup
  constexpr int VBUF = 4096;  // floats of one V buffer: [ks 4][xi 16][h 2][tile 32]
  __shared__ __attribute__((aligned(16))) float lds[16384];  // 64 KB: V double buffer (32 KB) / M exchange (64 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  // readfirstlane: tells hipcc the wave index is wave-uniform, otherwise every buffer load whose SGPR offset depends
  // on it is wrapped in a readfirstlane "waterfall" loop (measured: VALU time = 75 % of the MFMA time)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int tw = A.win >> 1, th = A.hin >> 1, ntiles = tw * th;
  const int b = blockIdx.z;
  // XCD-aware order over (tile block, cout block), as in conv_body
  const int ny = sessd_divup(A.cout_pad, 32);
  int bx, by;
  {
    const int total = gridDim.x, bid = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = bid & 7, loc = bid >> 3;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    bx = wgid / ny;
    by = wgid - bx * ny;
  }
  const int t_base = bx * TT, m_base = by * 32;
  const int in_plane = A.hin * A.win;
  const rsrc_t xr = make_rsrc(A.in + (size_t)b * A.cin * in_plane, (unsigned)A.cin * in_plane * 4u);
  const rsrc_t wr = make_rsrc(A.wpk, (unsigned)(A.cin >> 1) * 16u * 2u * A.cout_pad * 4u);

  // ---- transform role: lane = (tile j, channel parity h); 4 row offsets of its 4x4 patch
  const int t = t_base + j;
  const bool tlive = t < ntiles;
  const int ty = tlive ? t / tw : 0, tx = tlive ? t - (t / tw) * tw : 0;
  const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
  unsigned ro[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + r;
    // rows outside the image: out-of-range offset -> the hardware returns 0. Tiles in the first column start their
    // 16-byte row load at x = 0 (a negative offset would be range-checked away as a whole) and shift the components
    // in registers instead
    ro[r] = (tlive && y >= 0 && y < A.hin) ? (unsigned)((h * in_plane + y * A.win + max(x0, 0)) * 4) : SESSD_OOB;
  }
  const bool mask_l = (tx == 0), mask_r = (tx == tw - 1);
  const bool edge = __builtin_amdgcn_ballot_w64(tlive && (mask_l || mask_r)) != 0;
  // ---- GEMM role: wave owns xi = 4*wave .. 4*wave+3 ; A operand lane (cout i = j, channel parity h).
  // U is packed [cin/2][wave 4][h 2][cout_pad][xi_local 4]: one 16-byte load per lane and k-step
  const unsigned wo = (unsigned)(((wave * 2 + h) * A.cout_pad + m_base + j) * 16);
  const unsigned wstep = 4u * 2u * (unsigned)A.cout_pad * 16u;  // bytes per k-step
  const unsigned xstep = 2u * (unsigned)in_plane * 4u;          // bytes per k-step (2 channels)
  const int KP = A.cin >> 1, NR = KP / 4;                       // rounds of 4 k-steps

  f32x16 acc[4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  using f32x4v = __attribute__((ext_vector_type(4))) float;
  using f32x2v = __attribute__((ext_vector_type(2))) float;
  f32x4v pr[DEEP ? 2 : 1][4];  // patch rows of the k-step this wave transforms, per set
  f32x4v ua[DEEP ? 4 : 2][4];  // U operands of one round, per set: [ks_local] . xi_local

#define SESSD_WG_LOADP(PS, ROUND)                                                                  \
  {                                                                                                \
    const unsigned xs = (unsigned)(min((ROUND), NR - 1) * 4 + wave) * xstep;                       \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                  \
      pr[PS][r] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)ro[r], (int)xs, 0)); \
  }
#define SESSD_WG_LOADU(US, ROUND)                                                                  \
  {                                                                                                \
    const unsigned rb = (unsigned)(min((ROUND), NR - 1) * 4) * wstep;                              \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                               \
      ua[US][ks] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wo, (int)(rb + ks * wstep), 0)); \
  }
  // V of one round in LDS: [ks 4][xi 16][h 2][tile 32]
#define SESSD_WG_TRANSFORM(PS, BUF)                                                                \
  {                                                                                                \
    if (edge) { /* workgroup-uniform: only 1 tile block in 3 touches the left / right image border */ \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
        const f32x4v p = pr[PS][r];                                                                \
        pr[PS][r].x = mask_l ? 0.f : p.x; pr[PS][r].y = mask_l ? p.x : p.y;                        \
        pr[PS][r].z = mask_l ? p.y : p.z; pr[PS][r].w = mask_l ? p.z : (mask_r ? 0.f : p.w);       \
      }                                                                                            \
    }                                                                                              \
    /* rows: t = B^T d, two columns per packed add */                                              \
    f32x2v tl[4], tr[4];                                                                           \
    tl[0] = pr[PS][0].xy - pr[PS][2].xy; tr[0] = pr[PS][0].zw - pr[PS][2].zw;                      \
    tl[1] = pr[PS][1].xy + pr[PS][2].xy; tr[1] = pr[PS][1].zw + pr[PS][2].zw;                      \
    tl[2] = pr[PS][2].xy - pr[PS][1].xy; tr[2] = pr[PS][2].zw - pr[PS][1].zw;                      \
    tl[3] = pr[PS][1].xy - pr[PS][3].xy; tr[3] = pr[PS][1].zw - pr[PS][3].zw;                      \
    float* dst = &lds[(BUF)*VBUF + wave * 1024 + h * 32 + j];                                      \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                \
      dst[(a * 4 + 0) * 64] = tl[a].x - tr[a].x;                                                   \
      dst[(a * 4 + 1) * 64] = tl[a].y + tr[a].x;                                                   \
      dst[(a * 4 + 2) * 64] = tr[a].x - tl[a].y;                                                   \
      dst[(a * 4 + 3) * 64] = tl[a].y - tr[a].y;                                                   \
    }                                                                                              \
  }
#define SESSD_WG_MMA(US, BUF)                                                                      \
  {                                                                                                \
    const float* vb = &lds[(BUF)*VBUF + (wave * 4) * 64 + h * 32 + j];                             \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                               \
      _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                \
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[US][ks][x], vb[ks * 1024 + x * 64], acc[x], 0, 0, 0); \
  }

  if constexpr (!DEEP) {
    // round RR computes from V buffer PAR / U set PAR while the loads and the transform of round RR+1 proceed
#define SESSD_WG_ROUND(RR, PAR)                                                                    \
  {                                                                                                \
    SESSD_WG_LOADP(0, (RR) + 1)                                                                    \
    SESSD_WG_LOADU((PAR) ^ 1, (RR) + 1)                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_WG_MMA(PAR, PAR)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_WG_TRANSFORM(0, (PAR) ^ 1)                                                               \
    __syncthreads();                                                                               \
  }
    SESSD_WG_LOADP(0, 0)
    SESSD_WG_LOADU(0, 0)
    SESSD_WG_TRANSFORM(0, 0)
    __syncthreads();
    for (int R = 0; R < NR; R += 2) {
      SESSD_WG_ROUND(R, 0)
      if (R + 1 >= NR) break;
      SESSD_WG_ROUND(R + 1, 1)
    }
#undef SESSD_WG_ROUND
  } else {
    // round RR: MMA on V buffer P2 / U set U4; issue the loads of round RR+2 (patch set P2, U set (U4+2)%4);
    // transform the patches of round RR+1 (patch set P2^1) into V buffer P2^1
#define SESSD_WG_ROUND(RR, P2, U4)                                                                 \
  {                                                                                                \
    SESSD_WG_LOADP(P2, (RR) + 2)                                                                   \
    SESSD_WG_LOADU(((U4) + 2) & 3, (RR) + 2)                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_WG_MMA(U4, P2)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_WG_TRANSFORM((P2) ^ 1, (P2) ^ 1)                                                         \
    __syncthreads();                                                                               \
  }
    SESSD_WG_LOADP(0, 0)
    SESSD_WG_LOADU(0, 0)
    SESSD_WG_LOADP(1, 1)
    SESSD_WG_LOADU(1, 1)
    SESSD_WG_TRANSFORM(0, 0)
    __syncthreads();
    for (int R = 0; R < NR; R += 4) {
      SESSD_WG_ROUND(R, 0, 0)
      SESSD_WG_ROUND(R + 1, 1, 1)
      SESSD_WG_ROUND(R + 2, 0, 2)
      SESSD_WG_ROUND(R + 3, 1, 3)
    }
#undef SESSD_WG_ROUND
  }
#undef SESSD_WG_LOADP
#undef SESSD_WG_LOADU
#undef SESSD_WG_TRANSFORM
#undef SESSD_WG_MMA

  // ---- epilogue: M_xi[cout][tile] of all 16 xi through LDS, then Y = A^T M A per (cout, tile)
  // D layout: column = lane&31 (tile), row = (r&3) + 8*(r>>2) + 4*h (cout)
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
      lds[((wave * 4 + x) * 32 + co) * 32 + j] = acc[x][r];
    }
  __syncthreads();
  const size_t out_plane = (size_t)A.hout * A.wout;
  float* outb = A.out + (size_t)b * A.cout * out_plane;
  const float* resb = A.residual ? A.residual + (size_t)b * A.cout * out_plane : nullptr;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int pidx = tid + 256 * n;
    const int tl = pidx & 31, col = pidx >> 5;
    const int tt = t_base + tl, co = m_base + col;
    if (tt >= ntiles || co >= A.cout) continue;
    float m[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) m[x] = lds[(x * 32 + col) * 32 + tl];
    float q0[4], q1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      q0[c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
      q1[c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
    }
    float y[2][2];
    y[0][0] = q0[0] + q0[1] + q0[2]; y[0][1] = q0[1] - q0[2] - q0[3];
    y[1][0] = q1[0] + q1[1] + q1[2]; y[1][1] = q1[1] - q1[2] - q1[3];
    const int oy = 2 * (tt / tw), ox = 2 * (tt - (tt / tw) * tw);
    const float sc = A.scale ? A.scale[co] : 1.f, sh = A.shift ? A.shift[co] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const size_t o = (size_t)co * out_plane + (size_t)(oy + a) * A.wout + ox;
      float v0 = fmaf(y[a][0], sc, sh), v1 = fmaf(y[a][1], sc, sh);
      if (A.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      if (resb) { v0 += resb[o]; v1 += resb[o + 1]; }
      *reinterpret_cast<float2*>(outb + o) = make_float2(v0, v1);
    }
  }
}

template <int NTAPS, int CT, int PT, int WC, int WP, bool DEEP = false, bool LIST = false>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(ConvArgs A) {
  conv_body<NTAPS, CT, PT, WC, WP, DEEP, LIST>(A, LIST ? 0 : blockIdx.z);
}

// Up to four convolutions that share shapes but not weights / taps / output phase in ONE launch
// (the four output-parity classes of the stride-2 transposed conv): blockIdx.z = batch*4 + (3 - class). The classes
// have 1, 2, 2, 4 taps, i.e. 1x, 2x, 2x, 4x the work per workgroup: the heaviest class is dispatched first so that
// the tail of the launch is made of the light workgroups.
template <int NTAPS, int CT, int PT, int WC, int WP, bool DEEP = false>
__global__ __launch_bounds__(256) void conv2d_mfma4_kernel(ConvArgs4 A4) {
  conv_body<NTAPS, CT, PT, WC, WP, DEEP>(A4.c[3 - (blockIdx.z & 3)], blockIdx.z >> 2);
}

// Two transposed convs that read the SAME input (deconv_block_0 / deconv_block_1 of the SSFA neck, rpn_v1.py:224-226) in one
// launch: eight classes, c[2k] / c[2k+1] = class k of the first / second layer; blockIdx.z = batch*8 + (7 - index), heaviest first.
template <int NTAPS, int CT, int PT, int WC, int WP, bool DEEP = false, bool LIST = false>
__global__ __launch_bounds__(256) void conv2d_mfma8_kernel(ConvArgs8 A8) {
  conv_body<NTAPS, CT, PT, WC, WP, DEEP, LIST>(A8.c[7 - (blockIdx.z & 7)], LIST ? 0 : blockIdx.z >> 3);
}

// SSFA tail (rpn_v1.py:227-233): w0 = BN(conv1x1(x0)), w1 = BN(conv1x1(x1)) (C -> 1 channel, no ReLU),
// (s0, s1) = softmax(w0, w1), out = x0*s0 + x1*s1. Workgroup = 64 pixels x 4 channel quarters: each thread
// dots its quarter of the channels (coalesced across the 64 pixels of a wave), the four partial sums meet in
// LDS, then every thread blends its own quarter -- the second read of x0/x1 hits L2.
__global__ __launch_bounds__(256) void ssfa_fuse_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                         const float* __restrict__ w0, const float* __restrict__ w1,
                                                         float s0, float t0, float s1, float t1, int C, int npix,
                                                         float* __restrict__ out) {
  __shared__ float part[2][4][64];
  const int px = threadIdx.x & 63, cq = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + px;
  const int b = blockIdx.y;
  const int cper = C >> 2, c0 = cq * cper;
  const bool live = p < npix;
  const size_t base = (size_t)b * C * npix + (live ? p : 0);
  float a0 = 0.f, a1 = 0.f;
  if (live) {
#pragma unroll 8
    for (int c = c0; c < c0 + cper; ++c) {
      a0 = fmaf(x0[base + (size_t)c * npix], w0[c], a0);
      a1 = fmaf(x1[base + (size_t)c * npix], w1[c], a1);
    }
  }
  part[0][cq][px] = a0;
  part[1][cq][px] = a1;
  __syncthreads();
  // fixed summation order (quarter 0..3) so the result does not depend on which thread reads it
  a0 = ((part[0][0][px] + part[0][1][px]) + part[0][2][px]) + part[0][3][px];
  a1 = ((part[1][0][px] + part[1][1][px]) + part[1][2][px]) + part[1][3][px];
  a0 = fmaf(a0, s0, t0);
  a1 = fmaf(a1, s1, t1);
  const float m = fmaxf(a0, a1);
  const float e0 = expf(a0 - m), e1 = expf(a1 - m);
  const float inv = 1.f / (e0 + e1);
  const float p0 = e0 * inv, p1 = e1 * inv;
  if (!live) return;
#pragma unroll 8
  for (int c = c0; c < c0 + cper; ++c) {
    const size_t o = base + (size_t)c * npix;
    out[o] = x0[o] * p0 + x1[o] * p1;
  }
}

// The same tail FUSED with the 1x1 heads of the detection head (mg_head_sessd.py:217-230: box | cls | dir | iou = NOUT channels,
// bias, no activation): the blended value of a channel is multiplied into the NOUT head sums of its pixel while it is in a
// register, so the SSFA output (18 MB per frame) is neither written nor read back and one launch disappears. Thread = (pixel,
// channel quarter) as above; the head weights sit in LDS (all lanes of a wave read the same address: broadcast); the four
// quarter sums of a (pixel, head channel) meet in LDS and are added in quarter order. `out` may be null (inference).
template <int NOUT, int CPER>
__global__ __launch_bounds__(256, 2) void ssfa_fuse_head_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                              const float* __restrict__ w0, const float* __restrict__ w1,
                                                              float s0, float t0, float s1, float t1, int npix,
                                                              float* __restrict__ out, const float* __restrict__ hw,
                                                              const float* __restrict__ hb, float* __restrict__ hout,
                                                              float score_thresh, unsigned long long* __restrict__ keys,
                                                              int key_cap, int* __restrict__ key_count) {
  constexpr int C = 4 * CPER;
  __shared__ float part[2][4][64];
  __shared__ __attribute__((aligned(16))) float s_hw[NOUT * C];  // head weights
  __shared__ float s_acc[4 * NOUT * 64];                          // quarter sums of the head channels
  const int px = threadIdx.x & 63, cq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int p = blockIdx.x * 64 + px;
  const int b = blockIdx.y;
  const int c0 = cq * CPER;
  const bool live = p < npix;
  // buffer resources over this batch element's maps: lane offset = pixel, SGPR offset = channel plane (no 64-bit address per
  // channel in VGPRs); a dead lane reads / writes out of range
  const unsigned plane4 = (unsigned)npix * 4u, mbytes = (unsigned)C * plane4;
  const size_t boff = (size_t)b * C * npix;
  const rsrc_t r0 = make_rsrc(x0 + boff, mbytes), r1 = make_rsrc(x1 + boff, mbytes);
  const rsrc_t ro = make_rsrc(out ? out + boff : x0, out ? mbytes : 0u);
  const unsigned vp = live ? (unsigned)p * 4u : SESSD_OOB;
  // this thread's CPER channels of both maps: fetched once (all loads in flight together), used by the two dots and by the blend
  float v0[CPER], v1[CPER];
#pragma unroll
  for (int c = 0; c < CPER; ++c) {
    v0[c] = bufload(r0, vp, (unsigned)(c0 + c) * plane4);
    v1[c] = bufload(r1, vp, (unsigned)(c0 + c) * plane4);
  }
  for (int k = threadIdx.x; k < NOUT * C; k += 256) s_hw[k] = hw[k];
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int c = 0; c < CPER; ++c) {
    a0 = fmaf(v0[c], w0[c0 + c], a0);
    a1 = fmaf(v1[c], w1[c0 + c], a1);
  }
  part[0][cq][px] = a0;
  part[1][cq][px] = a1;
  __syncthreads();
  a0 = ((part[0][0][px] + part[0][1][px]) + part[0][2][px]) + part[0][3][px];
  a1 = ((part[1][0][px] + part[1][1][px]) + part[1][2][px]) + part[1][3][px];
  a0 = fmaf(a0, s0, t0);
  a1 = fmaf(a1, s1, t1);
  const float m = fmaxf(a0, a1);
  const float e0 = expf(a0 - m), e1 = expf(a1 - m);
  const float inv = 1.f / (e0 + e1);
  const float p0 = e0 * inv, p1 = e1 * inv;
  float acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
#pragma unroll
  for (int c = 0; c < CPER; c += 4) {
    float bl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bl[e] = v0[c + e] * p0 + v1[c + e] * p1;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bl[e]), ro, (int)vp, (int)((unsigned)(c0 + c + e) * plane4), 0);
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const float4 w = *reinterpret_cast<const float4*>(s_hw + o * C + c0 + c);
      acc[o] = fmaf(bl[0], w.x, acc[o]);
      acc[o] = fmaf(bl[1], w.y, acc[o]);
      acc[o] = fmaf(bl[2], w.z, acc[o]);
      acc[o] = fmaf(bl[3], w.w, acc[o]);
    }
    __builtin_amdgcn_sched_barrier(0);  // fully unrolled, hipcc would hoist all 8 x NOUT weight reads: 700 registers
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) s_acc[(cq * NOUT + o) * 64 + px] = acc[o];
  __syncthreads();
  if (!live) return;
  auto head_value = [&](int o) {  // quarter sums in quarter order + bias: the value stored for (pixel, head channel o)
    const float v = ((s_acc[(0 * NOUT + o) * 64 + px] + s_acc[(1 * NOUT + o) * 64 + px]) + s_acc[(2 * NOUT + o) * 64 + px]) +
                    s_acc[(3 * NOUT + o) * 64 + px];
    return v + (hb ? hb[o] : 0.f);
  };
  for (int o = cq; o < NOUT; o += 4) hout[((size_t)b * NOUT + o) * npix + p] = head_value(o);
  // The score filter of MultiGroupHead.predict (mg_head_sessd.py:956-972; postprocess.hip: score_filter_kernel) while the
  // logits are at hand: sigmoid(cls) >= thresh -> key (~rectified score | anchor id) appended to the frame's candidate list.
  // Planar head layout [box 14 | cls 2 | dir 4 | iou 2], two anchors per location; the same float operations on the same
  // values as the stand-alone kernel reads back from `hout`, so the keys are the same set.
  if (keys && cq == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float sg = 1.0f / (1.0f + expf(-head_value(14 + a)));
      if (sg >= score_thresh) {
        const float r = (head_value(20 + a) + 1.0f) * 0.5f;
        const float sc = sg * (r * r * r * r);
        const unsigned aid = (unsigned)(p * 2 + a);
        const unsigned long long key = ((unsigned long long)(~__float_as_uint(sc)) << 32) | aid;
        const int slot = atomicAdd(&key_count[b], 1);
        if (slot < key_cap) keys[(size_t)b * key_cap + slot] = key;
      }
    }
  }
}

template <int NTAPS, int CT, int PT, int WC, int WP, bool DEEP = false>
int launch_conv(const ConvArgs* A, int nconv, int batch, hipStream_t stream) {
  const int npix = A[0].ht * A[0].wt;
  dim3 grid(sessd_divup(npix, WP * PT * 32) * sessd_divup(A[0].cout_pad, WC * CT * 32), 1, batch * (nconv > 1 ? 4 : 1));
  if (nconv == 1) {
    SESSD_LAUNCH((conv2d_mfma_kernel<NTAPS, CT, PT, WC, WP, DEEP>), grid, dim3(256), 0, stream, A[0]);
  } else {
    ConvArgs4 A4;
    for (int i = 0; i < 4; ++i) A4.c[i] = A[i];
    SESSD_LAUNCH((conv2d_mfma4_kernel<NTAPS, CT, PT, WC, WP, DEEP>), grid, dim3(256), 0, stream, A4);
  }
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

template <int CT, int PT, int WC, int WP, bool DEEP>
int launch_conv8(const ConvArgs* A, int batch, hipStream_t stream) {
  const int npix = A[0].ht * A[0].wt;
  const bool list = A[0].tile_list != nullptr;   // the images of a list launch share the pixel axis
  dim3 grid(sessd_divup(npix * (list ? batch : 1), WP * PT * 32) * sessd_divup(A[0].cout_pad, WC * CT * 32), 1, (list ? 1 : batch) * 8);
  ConvArgs8 A8;
  for (int i = 0; i < 8; ++i) A8.c[i] = A[i];
  if (list)
    SESSD_LAUNCH((conv2d_mfma8_kernel<4, CT, PT, WC, WP, DEEP, true>), grid, dim3(256), 0, stream, A8);
  else
    SESSD_LAUNCH((conv2d_mfma8_kernel<4, CT, PT, WC, WP, DEEP>), grid, dim3(256), 0, stream, A8);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// one conv over a tile list (four effective taps: a 1x1 layer with its cin pairs grouped, or one class of a transposed conv)
template <int CT, int PT, int WC, int WP, bool DEEP>
int launch_conv_list(const ConvArgs& A, int batch, hipStream_t stream) {
  const int npix = A.ht * A.wt;
  dim3 grid(sessd_divup(npix * batch, WP * PT * 32) * sessd_divup(A.cout_pad, WC * CT * 32), 1, 1);
  SESSD_LAUNCH((conv2d_mfma_kernel<4, CT, PT, WC, WP, DEEP, true>), grid, dim3(256), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

template <int NTAPS>
int dispatch_tile(const ConvArgs* A, int nconv, int batch, int tile_cfg, hipStream_t stream) {
  switch (tile_cfg) {
    case 0: return launch_conv<NTAPS, 2, 2, 2, 2>(A, nconv, batch, stream);  // wave 64c x 64p, WG 128c x 128p
    case 1: return launch_conv<NTAPS, 2, 1, 2, 2>(A, nconv, batch, stream);  // wave 64c x 32p, WG 128c x 64p
    case 2: return launch_conv<NTAPS, 1, 2, 4, 1>(A, nconv, batch, stream);  // wave 32c x 64p, WG 128c x 64p
    case 3: return launch_conv<NTAPS, 1, 1, 4, 1>(A, nconv, batch, stream);  // wave 32c x 32p, WG 128c x 32p
    case 4: return launch_conv<NTAPS, 1, 1, 1, 4>(A, nconv, batch, stream);  // wave 32c x 32p, WG 32c x 128p (small cout)
    case 5: return launch_conv<NTAPS, 2, 1, 4, 1>(A, nconv, batch, stream);  // wave 64c x 32p, WG 256c x 32p
    case 6: return launch_conv<NTAPS, 2, 1, 1, 4>(A, nconv, batch, stream);  // wave 64c x 32p, WG 64c x 128p (weights shared by the 4 waves)
    case 7: return launch_conv<NTAPS, 2, 2, 1, 4>(A, nconv, batch, stream);  // wave 64c x 64p, WG 64c x 256p
    case 8: return launch_conv<NTAPS, 4, 1, 1, 4>(A, nconv, batch, stream);  // wave 128c x 32p, WG 128c x 128p
    case 11: return launch_conv<NTAPS, 1, 1, 1, 4, true>(A, nconv, batch, stream);  // cfg 4, 2-step look-ahead
    case 12: return launch_conv<NTAPS, 1, 1, 4, 1, true>(A, nconv, batch, stream);  // cfg 3, 2-step look-ahead
    case 13: if (nconv == 1) return launch_conv<NTAPS, 1, 2, 4, 1, true>(A, nconv, batch, stream); return SESSD_EINVAL;  // cfg 2, 2-step look-ahead
    default: return SESSD_EINVAL;
  }
}

// Fill one ConvArgs; few-tap convolutions are regrouped into NTAPS_eff "virtual taps" (several cin pairs per
// k-step) so that every k-step issues a full batch of loads ahead of its MFMAs. Returns the effective tap count.
// the epilogue addresses a batch element's output through a 32-bit buffer offset
inline bool out_fits(int cout, int hout, int wout) { return (long long)cout * hout * wout * 4 < 0x7fffffffLL; }

int fill_args(ConvArgs& A, const float* in, int cin, int hin, int win, const float* wpk, int ntaps, const int* dy,
              const int* dx, int in_mul, int tile_h, int tile_w, float* out, int cout, int hout, int wout, int out_mul,
              int py, int px, const float* scale, const float* shift, int relu, const float* residual) {
  A.in = in; A.wpk = wpk; A.out = out; A.scale = scale; A.shift = shift; A.residual = residual;
  A.cin = cin; A.hin = hin; A.win = win;
  A.cout = cout; A.cout_pad = sessd_divup(cout, 32) * 32; A.hout = hout; A.wout = wout;
  A.ht = tile_h; A.wt = tile_w;
  A.in_mul = in_mul; A.out_mul = out_mul; A.out_py = py; A.out_px = px;
  A.relu = relu;
  A.tile_list = nullptr; A.n_list = nullptr; A.list_cap = 0; A.ntiles2 = 1; A.tw2 = 1; A.lbatch = 1;
  int group = 1;
  const int pairs = cin / 2;
  if (ntaps == 1 && pairs % 4 == 0) group = 4;
  else if (ntaps == 2 && pairs % 2 == 0) group = 2;
  A.cgroup = group;
  const int eff = ntaps * group;
  // packed layout [pair][tap][2][cout_pad]: virtual tap v of a group = (pair displacement v / ntaps, tap v % ntaps)
  for (int v = 0; v < 9; ++v) {
    const int t = v < eff ? v % ntaps : 0, g = v < eff ? v / ntaps : 0;
    A.dy[v] = dy[t]; A.dx[v] = dx[t]; A.dc[v] = g;
  }
  return eff;
}

}  // namespace

extern "C" {

// Generic conv launcher. taps_dy/taps_dx: host int[ntaps] input offsets; wpk packed [cin/2][ntaps][2][cout_pad]
// (cout_pad = cout rounded up to 32, padding columns zero). tile_cfg selects the wave/workgroup tiling (0..5).
int sessd_conv2d_mfma(const float* in, int batch, int cin, int hin, int win, const float* wpk, int ntaps,
                      const int* taps_dy, const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out, int cout,
                      int hout, int wout, int out_mul, int out_py, int out_px, const float* scale, const float* shift,
                      int relu, const float* residual, int tile_cfg, hipStream_t stream) {
  if (cin % 2 || ntaps < 1 || ntaps > 9 || batch < 1 || cout < 1 || !out_fits(cout, hout, wout)) return SESSD_EINVAL;
  ConvArgs A;
  const int eff = fill_args(A, in, cin, hin, win, wpk, ntaps, taps_dy, taps_dx, in_mul, tile_h, tile_w, out, cout, hout,
                            wout, out_mul, out_py, out_px, scale, shift, relu, residual);
  if (tile_cfg == 10) {  // activation-stationary LDS variant (3x3, stride 1, cin % 16 == 0)
    if (eff != 9 || in_mul != 1 || out_mul != 1 || cin % 16 || hin != tile_h || win != tile_w) return SESSD_EINVAL;
    dim3 grid(sessd_divup(tile_h * tile_w, 32), sessd_divup(A.cout_pad, 128), batch);
    SESSD_LAUNCH(conv3x3s1_lds_kernel, grid, dim3(256), 0, stream, A);
    SESSD_CHECK_LAUNCH();
    return SESSD_OK;
  }
  switch (eff) {
    case 1: return dispatch_tile<1>(&A, 1, batch, tile_cfg, stream);
    case 2: return dispatch_tile<2>(&A, 1, batch, tile_cfg, stream);
    case 4: return dispatch_tile<4>(&A, 1, batch, tile_cfg, stream);
    case 9: return dispatch_tile<9>(&A, 1, batch, tile_cfg, stream);
    default: return SESSD_EINVAL;
  }
}

// Fused Winograd F(2x2,3x3) for Conv2d(cin, cout, 3, stride 1, padding 1): upk = packed U = G g G^T as
// [cin/2][4][2][cout_pad][4] (xi = 4*row + col of the 4x4 transform domain, row-major outer/inner). Even H, W;
// cin % 8 == 0. variant 0 / 1 = loads one / two rounds ahead of their use.
int sessd_conv3x3_winograd(const float* in, int batch, int cin, int h, int w, const float* upk, float* out, int cout,
                           const float* scale, const float* shift, int relu, const float* residual, int variant,
                           hipStream_t stream) {
  if (cin % 8 || (h & 1) || (w & 1) || batch < 1 || cout < 1) return SESSD_EINVAL;
  if (variant < 0 || variant > 1 || (variant == 1 && cin % 32)) return SESSD_EINVAL;
  ConvArgs A;
  const int z9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  fill_args(A, in, cin, h, w, upk, 9, z9, z9, 1, h, w, out, cout, h, w, 1, 0, 0, scale, shift, relu, residual);
  const int ntiles = (h / 2) * (w / 2);
  dim3 grid(sessd_divup(ntiles, 32) * sessd_divup(A.cout_pad, 32), 1, batch);
  if (variant == 1)
    SESSD_LAUNCH(conv3x3s1_winograd_kernel<true>, grid, dim3(256), 0, stream, A);
  else
    SESSD_LAUNCH(conv3x3s1_winograd_kernel<false>, grid, dim3(256), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// ---- weight packing on the device (the training step re-packs every dense conv weight three times per iteration: teacher
// forward, student forward, student data gradient; as torch permute / stack / einsum chains that was ~2 ms of small launches)
}  // extern "C"
namespace {
struct PackArgs {
  const float* w;
  long long so, sc;     // element strides of the (virtual) output / input channel in w
  int tap_off[16];      // element offset of tap t inside a (channel, channel) filter
  int co, ci, nt, cp;
};
// out [ci/2][nt][2][cp]: out[((kp*nt + t)*2 + h)*cp + o] = w[o*so + (2kp+h)*sc + tap_off[t]], zero for o >= co
__global__ __launch_bounds__(256) void pack_taps_kernel(PackArgs A, float* __restrict__ out) {
  const size_t total = (size_t)(A.ci >> 1) * A.nt * 2 * A.cp;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int o = (int)(idx % A.cp);
  size_t r = idx / A.cp;
  const int h = (int)(r & 1); r >>= 1;
  const int t = (int)(r % A.nt);
  const int kp = (int)(r / A.nt);
  out[idx] = o < A.co ? A.w[(long long)o * A.so + (long long)(2 * kp + h) * A.sc + A.tap_off[t]] : 0.f;
}
// U = G g G^T (float64 arithmetic, rounded once) of the 3x3 filter of (o, c), flipped when flip != 0, written in
// layout 0: sessd_conv3x3_winograd      [ci/2][xi/4][h][cp32][xi%4]
// layout 1: sessd_conv3x3_winograd_sk shape 0  [ceil(co/128)][ci/2][8][2][32][4][2]
// layout 2: sessd_conv3x3_winograd_sk shape 1  [ceil(co/64)][ci/2][4][2][32][2][4]
// layout 3: sessd_conv3x3_winograd_sk shape 2  [ceil(co/128)][ci/2][wave 4][2][32][xi 16]
__device__ __forceinline__ void winograd_pack_body(const PackArgs& A, int flip, int layout, float* __restrict__ out, size_t idx) {
  const int cpad = layout == 0 ? A.cp : ((layout == 1 || layout == 3) ? (A.co + 127) / 128 * 128 : (A.co + 63) / 64 * 64);
  if (idx >= (size_t)cpad * A.ci) return;
  const int o = (int)(idx % cpad), c = (int)(idx / cpad);
  double g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int t = flip ? 8 - (3 * a + b) : 3 * a + b;
      g[a][b] = o < A.co ? (double)A.w[(long long)o * A.so + (long long)c * A.sc + t] : 0.0;
    }
  // rows of G: [1,0,0], [.5,.5,.5], [.5,-.5,.5], [0,0,1]
  double t0[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t0[0][b] = g[0][b];
    t0[1][b] = 0.5 * g[0][b] + 0.5 * g[1][b] + 0.5 * g[2][b];
    t0[2][b] = 0.5 * g[0][b] - 0.5 * g[1][b] + 0.5 * g[2][b];
    t0[3][b] = g[2][b];
  }
  const int kp = c >> 1, h = c & 1;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const double u[4] = {t0[a][0], 0.5 * t0[a][0] + 0.5 * t0[a][1] + 0.5 * t0[a][2], 0.5 * t0[a][0] - 0.5 * t0[a][1] + 0.5 * t0[a][2],
                         t0[a][2]};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int xi = a * 4 + b;
      size_t dst;
      if (layout == 0) {
        dst = ((((size_t)kp * 4 + (xi >> 2)) * 2 + h) * cpad + o) * 4 + (xi & 3);
      } else if (layout == 1) {
        const int grp = o >> 7, cb = (o >> 5) & 3, j = o & 31, wave = xi >> 1, xl = xi & 1;
        dst = ((((((size_t)grp * (A.ci >> 1) + kp) * 8 + wave) * 2 + h) * 32 + j) * 4 + cb) * 2 + xl;
      } else if (layout == 3) {
        const int grp = o >> 7, wave = (o >> 5) & 3, j = o & 31;
        dst = (((((size_t)grp * (A.ci >> 1) + kp) * 4 + wave) * 2 + h) * 32 + j) * 16 + xi;
      } else {
        const int grp = o >> 6, cb = (o >> 5) & 1, j = o & 31, wave = xi >> 2, xl = xi & 3;
        dst = ((((((size_t)grp * (A.ci >> 1) + kp) * 4 + wave) * 2 + h) * 32 + j) * 2 + cb) * 4 + xl;
      }
      out[dst] = (float)u[b];
    }
  }
}
__global__ __launch_bounds__(256) void winograd_pack_kernel(PackArgs A, int flip, int layout, float* __restrict__ out) {
  winograd_pack_body(A, flip, layout, out, (size_t)blockIdx.x * 256 + threadIdx.x);
}
// All weight packings of a model pass in ONE launch: block -> job by binary search over the jobs' first blocks, then the body
// of pack_taps_kernel / winograd_pack_kernel on the job's arguments (read from the device table).
__global__ __launch_bounds__(256) void dense_pack_batch_kernel(const sessd_dense_pack_job_t* __restrict__ jobs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const sessd_dense_pack_job_t* J = jobs + lo;
  const size_t idx = (size_t)((int)blockIdx.x - J->block_start) * 256 + threadIdx.x;
  const int co = J->cout, ci = J->cin, cp = sessd_divup(co, 32) * 32;
  const float* __restrict__ w = J->w;
  float* __restrict__ out = J->out;
  const long long so = J->out_stride, sc = J->in_stride;
  if (J->kind == 0) {
    const int nt = J->ntaps;
    const size_t total = (size_t)(ci >> 1) * nt * 2 * cp;
    if (idx >= total) return;
    const int o = (int)(idx % cp);
    size_t r = idx / cp;
    const int h = (int)(r & 1); r >>= 1;
    const int t = (int)(r % nt);
    const int kp = (int)(r / nt);
    out[idx] = o < co ? w[(long long)o * so + (long long)(2 * kp + h) * sc + J->tap_off[t]] : 0.f;
    return;
  }
  PackArgs A;
  A.w = w; A.so = so; A.sc = sc; A.co = co; A.ci = ci; A.nt = 9; A.cp = cp;
  winograd_pack_body(A, J->flip, J->layout, out, idx);
}
}  // namespace
extern "C" {

int sessd_dense_pack_batch(const sessd_dense_pack_job_t* jobs_dev, int n_jobs, int total_blocks, hipStream_t stream) {
  if (!jobs_dev || n_jobs < 1 || total_blocks < 1) return SESSD_EINVAL;
  SESSD_LAUNCH(dense_pack_batch_kernel, dim3(total_blocks), dim3(256), 0, stream, jobs_dev, n_jobs);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// Pack a conv weight for sessd_conv2d_mfma / sessd_deconv2d_s2_mfma: out [cin/2][ntaps][2][cout_pad32] with
// out[kp][t][h][o] = w[o * out_stride + (2 kp + h) * in_stride + tap_offsets[t]] (element strides / offsets into w, so that a
// transposed, flipped or tap-selected view of the stored weight needs no intermediate tensor). ntaps <= 16.
int sessd_conv2d_pack_taps(const float* w, long long out_stride, long long in_stride, const int* tap_offsets, int ntaps, int cout,
                           int cin, float* out, hipStream_t stream) {
  if (ntaps < 1 || ntaps > 16 || cout < 1 || cin < 2 || (cin & 1)) return SESSD_EINVAL;
  PackArgs A;
  A.w = w; A.so = out_stride; A.sc = in_stride; A.co = cout; A.ci = cin; A.nt = ntaps; A.cp = sessd_divup(cout, 32) * 32;
  for (int t = 0; t < 16; ++t) A.tap_off[t] = t < ntaps ? tap_offsets[t] : 0;
  const size_t total = (size_t)(cin >> 1) * ntaps * 2 * A.cp;
  SESSD_LAUNCH(pack_taps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, A, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// U = G g G^T of a 3x3 weight viewed through (out_stride, in_stride, flip) as above, in the layout of sessd_conv3x3_winograd
// (layout 0) or sessd_conv3x3_winograd_sk shape 0 / 1 (layout 1 / 2); `out` must hold the padded layout (all of it is written).
int sessd_conv3x3_winograd_pack(const float* w, long long out_stride, long long in_stride, int flip, int cout, int cin, int layout,
                                float* out, hipStream_t stream) {
  if (cout < 1 || cin < 2 || (cin & 1) || layout < 0 || layout > 3) return SESSD_EINVAL;
  PackArgs A;
  A.w = w; A.so = out_stride; A.sc = in_stride; A.co = cout; A.ci = cin; A.nt = 9; A.cp = sessd_divup(cout, 32) * 32;
  for (int t = 0; t < 16; ++t) A.tap_off[t] = 0;
  const int cpad = layout == 0 ? A.cp : ((layout == 1 || layout == 3) ? sessd_divup(cout, 128) * 128 : sessd_divup(cout, 64) * 64);
  const size_t total = (size_t)cpad * cin;
  SESSD_LAUNCH(winograd_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, A, flip, layout, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// tile_cfg 40 .. 42: both px classes of a row parity in every wave, whole-line stores (dense_deconv_pair.hip)
}  // extern "C"
int sessd_deconv_pair_launch(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4, float* out, int cout,
                             const float* scale, const float* shift, int relu, const float* residual, int variant,
                             hipStream_t stream);
extern "C" {

// ConvTranspose2d(cin, cout, 3, stride 2, padding 1, output_padding 1) as ONE launch over its four output-parity
// classes. wpk4[c], ntaps4[c], taps_dy4/taps_dx4 (4 x 4 ints, row c = class c) in class order (py,px) =
// (0,0),(0,1),(1,0),(1,1) with 1,2,2,4 taps; input (B,cin,hin,win) -> output (B,cout,2*hin,2*win).
int sessd_deconv2d_s2_mfma(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4,
                           const int* ntaps4, const int* taps_dy4, const int* taps_dx4, float* out, int cout,
                           const float* scale, const float* shift, int relu, const float* residual, int tile_cfg,
                           hipStream_t stream) {
  if (cin % 8 || batch < 1 || cout < 1 || !out_fits(cout, 2 * hin, 2 * win)) return SESSD_EINVAL;
  if (tile_cfg >= 40 && tile_cfg <= 42) {
    // the paired kernel hard-codes the class tap tables of ops.pack_deconv2d_s2: refuse anything else
    static const int nt[4] = {1, 2, 2, 4};
    static const int dy[16] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0};
    static const int dx[16] = {0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0};
    for (int c = 0; c < 4; ++c) {
      if (ntaps4[c] != nt[c]) return SESSD_EINVAL;
      for (int t = 0; t < nt[c]; ++t)
        if (taps_dy4[4 * c + t] != dy[4 * c + t] || taps_dx4[4 * c + t] != dx[4 * c + t]) return SESSD_EINVAL;
    }
    return sessd_deconv_pair_launch(in, batch, cin, hin, win, wpk4, out, cout, scale, shift, relu, residual, tile_cfg - 40, stream);
  }
  ConvArgs A[4];
  for (int c = 0; c < 4; ++c) {
    const int eff = fill_args(A[c], in, cin, hin, win, wpk4[c], ntaps4[c], taps_dy4 + 4 * c, taps_dx4 + 4 * c, 1, hin, win,
                              out, cout, 2 * hin, 2 * win, 2, c >> 1, c & 1, scale, shift, relu, residual);
    if (eff != 4) return SESSD_EINVAL;
  }
  return dispatch_tile<4>(A, 4, batch, tile_cfg, stream);
}

// Two ConvTranspose2d(cin, cout, 3, 2, 1, 1) layers applied to the SAME input (deconv_block_0 / deconv_block_1 of the SSFA neck)
// as one launch over their 2 x 4 parity classes: per layer wpk4 / out / scale / shift / residual as in sessd_deconv2d_s2_mfma,
// shared tap tables. tile_cfg 3, 4, 11 or 12. Per class the code of the single-layer launch: the same bits.
int sessd_deconv2d_s2_mfma_pair(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4_a,
                                const float* const* wpk4_b, const int* ntaps4, const int* taps_dy4, const int* taps_dx4,
                                float* out_a, float* out_b, int cout, const float* scale_a, const float* shift_a,
                                const float* scale_b, const float* shift_b, int relu, const float* residual_a,
                                const float* residual_b, int tile_cfg, hipStream_t stream) {
  if (cin % 8 || batch < 1 || cout < 1 || !out_fits(cout, 2 * hin, 2 * win)) return SESSD_EINVAL;
  ConvArgs A[8];
  for (int c = 0; c < 4; ++c)
    for (int l = 0; l < 2; ++l) {
      const int eff = fill_args(A[2 * c + l], in, cin, hin, win, (l ? wpk4_b : wpk4_a)[c], ntaps4[c], taps_dy4 + 4 * c, taps_dx4 + 4 * c,
                                1, hin, win, l ? out_b : out_a, cout, 2 * hin, 2 * win, 2, c >> 1, c & 1, l ? scale_b : scale_a,
                                l ? shift_b : shift_a, relu, l ? residual_b : residual_a);
      if (eff != 4) return SESSD_EINVAL;
    }
  switch (tile_cfg) {
    case 3: return launch_conv8<1, 1, 4, 1, false>(A, batch, stream);
    case 4: return launch_conv8<1, 1, 1, 4, false>(A, batch, stream);
    case 11: return launch_conv8<1, 1, 1, 4, true>(A, batch, stream);
    case 12: return launch_conv8<1, 1, 4, 1, true>(A, batch, stream);
    default: return SESSD_EINVAL;
  }
}

// ACTIVE-TILE mode of the two launches above (csrc/dense_active.hip; rpn_v1.py:163-199 on maps that are a per-channel constant
// away from the sparse sites): only the 2x2 tiles of the TILE SPACE listed in tile_list[0 .. min(*n_list, list_cap)) (entries
// image * (tile_h/2 * tile_w/2) + tile; count on the device) are computed -- for the transposed convs the tile space is the INPUT
// map, a listed tile gives a 4x4 block of output pixels --, the other output pixels are left alone. The launch is sized for the
// whole map; workgroups beyond the count leave at once. Even tile_h / tile_w, the whole batch inside 32-bit buffer offsets.
// Per computed pixel the code of the plain launch: the same bits.
int sessd_conv2d_mfma_active(const float* in, int batch, int cin, int hin, int win, const float* wpk, int ntaps, const int* taps_dy,
                             const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out, int cout, int hout, int wout,
                             int out_mul, int out_py, int out_px, const float* scale, const float* shift, int relu,
                             const float* residual, const int32_t* tile_list, const int32_t* n_list, int list_cap, int tile_cfg,
                             hipStream_t stream) {
  if (cin % 2 || ntaps < 1 || ntaps > 9 || batch < 1 || cout < 1 || !tile_list || !n_list || list_cap < 1 || (tile_h & 1) || (tile_w & 1))
    return SESSD_EINVAL;
  if ((long long)batch * cout * hout * wout * 4 >= 0x7fffffffLL || (long long)batch * cin * hin * win * 4 >= 0x7fffffffLL) return SESSD_EINVAL;
  ConvArgs A;
  const int eff = fill_args(A, in, cin, hin, win, wpk, ntaps, taps_dy, taps_dx, in_mul, tile_h, tile_w, out, cout, hout, wout, out_mul,
                            out_py, out_px, scale, shift, relu, residual);
  if (eff != 4) return SESSD_EINVAL;   // 1x1 layers with cin % 8 == 0 (four cin pairs per k-step) and 4-tap classes
  A.tile_list = tile_list; A.n_list = n_list; A.list_cap = list_cap; A.ntiles2 = (tile_h / 2) * (tile_w / 2); A.tw2 = tile_w / 2;
  A.lbatch = batch;
  switch (tile_cfg) {
    case 3: return launch_conv_list<1, 1, 4, 1, false>(A, batch, stream);
    case 4: return launch_conv_list<1, 1, 1, 4, false>(A, batch, stream);
    case 11: return launch_conv_list<1, 1, 1, 4, true>(A, batch, stream);
    case 12: return launch_conv_list<1, 1, 4, 1, true>(A, batch, stream);
    default: return SESSD_EINVAL;
  }
}

int sessd_deconv2d_s2_mfma_pair_active(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4_a,
                                       const float* const* wpk4_b, const int* ntaps4, const int* taps_dy4, const int* taps_dx4,
                                       float* out_a, float* out_b, int cout, const float* scale_a, const float* shift_a,
                                       const float* scale_b, const float* shift_b, int relu, const float* residual_a,
                                       const float* residual_b, const int32_t* tile_list, const int32_t* n_list, int list_cap,
                                       int tile_cfg, hipStream_t stream) {
  if (cin % 8 || batch < 1 || cout < 1 || !tile_list || !n_list || list_cap < 1 || (hin & 1) || (win & 1)) return SESSD_EINVAL;
  if ((long long)batch * cout * 4 * hin * win * 4 >= 0x7fffffffLL || (long long)batch * cin * hin * win * 4 >= 0x7fffffffLL) return SESSD_EINVAL;
  ConvArgs A[8];
  for (int c = 0; c < 4; ++c)
    for (int l = 0; l < 2; ++l) {
      ConvArgs& C = A[2 * c + l];
      const int eff = fill_args(C, in, cin, hin, win, (l ? wpk4_b : wpk4_a)[c], ntaps4[c], taps_dy4 + 4 * c, taps_dx4 + 4 * c, 1, hin, win,
                                l ? out_b : out_a, cout, 2 * hin, 2 * win, 2, c >> 1, c & 1, l ? scale_b : scale_a,
                                l ? shift_b : shift_a, relu, l ? residual_b : residual_a);
      if (eff != 4) return SESSD_EINVAL;
      C.tile_list = tile_list; C.n_list = n_list; C.list_cap = list_cap; C.ntiles2 = (hin / 2) * (win / 2); C.tw2 = win / 2;
      C.lbatch = batch;
    }
  switch (tile_cfg) {
    case 3: return launch_conv8<1, 1, 4, 1, false>(A, batch, stream);
    case 4: return launch_conv8<1, 1, 1, 4, false>(A, batch, stream);
    case 11: return launch_conv8<1, 1, 1, 4, true>(A, batch, stream);
    case 12: return launch_conv8<1, 1, 4, 1, true>(A, batch, stream);
    default: return SESSD_EINVAL;
  }
}

// SSFA fusion tail: x0, x1, out are (B, C, H, W); w0, w1 the (C,) 1x1 conv weights; (s, t) the folded
// single-channel BatchNorm of each weight branch.
int sessd_ssfa_fuse(const float* x0, const float* x1, const float* w0, const float* w1, float bn_scale0,
                    float bn_shift0, float bn_scale1, float bn_shift1, int batch, int channels, int num_pixels,
                    float* out, hipStream_t stream) {
  if (batch < 1 || channels < 1 || num_pixels < 1) return SESSD_EINVAL;
  if (channels % 4) return SESSD_EINVAL;
  SESSD_LAUNCH(ssfa_fuse_kernel, dim3(sessd_divup(num_pixels, 64), batch), dim3(256), 0, stream, x0, x1, w0, w1,
                     bn_scale0, bn_shift0, bn_scale1, bn_shift1, channels, num_pixels, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}


// rpn_v1.py:227-233 + mg_head_sessd.py:217-230 in one launch: the SSFA fusion tail with the four 1x1 heads applied to its result
// while it is in registers. head_w (nout, channels) row-major = the concatenated conv weights, head_b (nout) or null,
// head_out (B, nout, num_pixels) planar. out (B, C, num_pixels) receives the SSFA output when not null. nout == 22 (the
// single-task car head: 14 box + 2 cls + 4 dir + 2 iou), channels 128 (the SSFA neck) or 64.
// keys != NULL: the launch also runs the score filter of predict (score_thresh on sigmoid(cls), IoU-rectified score) and
// appends the candidates' 64-bit keys (~score bits << 32 | anchor id, anchor id = 2 * pixel + a) to keys[b * key_cap ..] with
// key_count[b] (zeroed by the caller) counting them -- what sessd_predict_fused takes as ext_keys / ext_key_count.
int sessd_ssfa_fuse_head_keys(const float* x0, const float* x1, const float* w0, const float* w1, float bn_scale0, float bn_shift0,
                              float bn_scale1, float bn_shift1, int batch, int channels, int num_pixels, float* out,
                              const float* head_w, const float* head_b, int nout, float* head_out, float score_thresh,
                              unsigned long long* keys, int key_cap, int* key_count, hipStream_t stream) {
  if (batch < 1 || (channels != 128 && channels != 64) || num_pixels < 1 || nout != 22) return SESSD_EINVAL;
  if ((long long)channels * num_pixels * 4 >= 0x7fffffffLL) return SESSD_EINVAL;  // 32-bit buffer offsets per batch element
  if ((keys == nullptr) != (key_count == nullptr) || (keys && key_cap < 1)) return SESSD_EINVAL;
  const dim3 grid(sessd_divup(num_pixels, 64), batch);
  if (channels == 128)
    SESSD_LAUNCH((ssfa_fuse_head_kernel<22, 32>), grid, dim3(256), 0, stream, x0, x1, w0, w1, bn_scale0, bn_shift0, bn_scale1,
                 bn_shift1, num_pixels, out, head_w, head_b, head_out, score_thresh, keys, key_cap, key_count);
  else
    SESSD_LAUNCH((ssfa_fuse_head_kernel<22, 16>), grid, dim3(256), 0, stream, x0, x1, w0, w1, bn_scale0, bn_shift0, bn_scale1,
                 bn_shift1, num_pixels, out, head_w, head_b, head_out, score_thresh, keys, key_cap, key_count);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_ssfa_fuse_head(const float* x0, const float* x1, const float* w0, const float* w1, float bn_scale0, float bn_shift0,
                         float bn_scale1, float bn_shift1, int batch, int channels, int num_pixels, float* out,
                         const float* head_w, const float* head_b, int nout, float* head_out, hipStream_t stream) {
  return sessd_ssfa_fuse_head_keys(x0, x1, w0, w1, bn_scale0, bn_shift0, bn_scale1, bn_shift1, batch, channels, num_pixels, out,
                                   head_w, head_b, nout, head_out, 0.f, nullptr, 0, nullptr, stream);
}

}  // extern "C"
