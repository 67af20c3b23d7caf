// Dense BEV convolutions that are NOT 3x3 stride 1 (the stride-2 conv, the 1x1 convs and the four output-parity classes of the
// stride-2 transposed convs of the SSFA neck, det3d/models/necks/rpn_v1.py:150-210) as an LDS-tiled implicit GEMM on the f32
// matrix cores, decomposed "stream-K" (tile_cfg 30). Second generation of conv2d_mfma_kernel (dense_conv.hip), written after its
// counters: that kernel feeds every MFMA with two global dword loads straight into registers, and at ~30 B/clk/CU of L1 fill
// rate the chip cannot deliver more than ~45 % of the f32 MFMA peak that way; and a batch-1 layer is 138 .. 276 workgroup tiles
// on 256 CUs -- one and a bit waves of workgroups.
//
//   D[cout][pixel] += W[cout][k] X[k][pixel],  k = (input channel, tap)
//   unit      = 128 couts x 128 output pixels (of one batch element and one output-parity class) x all k
//   round     = 16 input channels of one tap ("chunk"); a unit has ntaps * cin/16 rounds, channel block major, tap minor
//   workgroup = 4 waves as 2 x 2, each 64 couts x 64 pixels = 2 x 2 MFMA tiles (v_mfma_f32_32x32x2_f32), 32 MFMAs per round
//   operands  : both through LDS, double buffered. The chunk of round r+2 is fetched to registers while round r is multiplied
//               (W: pre-packed in exactly the LDS image, two 16-B loads per thread; X: thread = (pixel, channel half), 8 dword
//               loads, coalesced over the pixels, zero padding by out-of-range buffer offsets) and written to the buffer round r
//               has left; the fragments of round r+1 are read (8 x ds_read_b128 per lane, conflict free: the image is
//               [channel half h][quad q][row 128][4] with MFMA step s = 4q + e multiplying channels s and 8 + s) into the
//               second fragment set before the MFMAs of round r start. One barrier per round; one LDS dword per MFMA
//               instead of two global ones.
//   stream-K  : `workgroups` persistent workgroups; the list of all rounds of all units (classes with more taps first) is cut
//               into equal contiguous shares -- the operand pipeline runs straight through unit boundaries. A unit cut by a
//               share boundary: every part writes its partial tile to its scratch slot (system-scope write-through), counts
//               itself on the unit's counter, and the part that counts last adds all parts in share order and finishes the
//               unit. No workgroup waits for another; the summation order is fixed.
// Numerics: fmaf chains per (share, MFMA step), parts added in share order -- exact float32 arithmetic, deterministic for a
// fixed (layer shape, batch, workgroups); NOT the single chain of the direct kernel (last-bit differences).
#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;
typedef unsigned int u32x4g __attribute__((__vector_size__(16)));  // the b128 buffer builtins' own type

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
#define SESSD_OOB 0x80000000u
#define SESSD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SESSD_SYSTEM_SCOPE 17  // sc0 | sc1: write-through to / read from memory, past the per-XCD L2

constexpr int CSK_SLOT_BYTES = 128 * 128 * 4;  // one partial tile
constexpr int CSK_CHUNK_FLOATS = 2048;         // 128 rows x 16 channels

struct SkClass {
  const float* wpk;  // [cout groups][cin/16][ntaps][2048]
  int ntaps, py, px;
  int rpu;           // rounds per unit = ntaps * cin/16
  int r_begin;       // first round of the class's unit inside a group (the units of all classes for one pixel tile and cout group)
  int u_begin;       // unused
  int dy[9], dx[9];
};

struct SkArgs {
  const float* in;        // (B, cin, hin, win)
  float* out;             // (B, cout, hout, wout)
  const float* scale;
  const float* shift;
  const float* residual;
  float* scratch;         // [2 * workgroups][128 x 128]
  unsigned* counters;     // [units], zero between launches
  int cin, hin, win, cout, hout, wout, wt, in_mul, out_mul, relu;
  int npix, ptiles, cgroups, ncb, batch, nclass, total_rounds;
  int rpg;                // rounds per group = cin/16 * sum of the classes' taps
  // active-tile mode (LIST kernel): entries image * (tile_h/2 * tile_w/2) + 2x2 tile of the TILE-SPACE map, count on the device
  const int* tile_list;
  const int* n_list;
  int list_cap, min_rounds, ntiles2, tw2;
  SkClass cls[4];
};

template <bool LIST>
__global__ __launch_bounds__(256) void conv2d_sk_kernel(SkArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * CSK_CHUNK_FLOATS];  // [buffer][W image | X image]
  __shared__ unsigned xo_tab[9 * 256];  // this thread's input offset per tap of the unit being loaded (thread-private rows)
  __shared__ int s_flag;
  const __attribute__((address_space(4))) SkArgs* Kp =
      (const __attribute__((address_space(4))) SkArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int wc = wave & 1, wp = wave >> 1;
  const int hh = wave >> 1;  // loader role: channel half of the chunk (threads 0..127 / 128..255)
  const int lp = tid & 127;  // loader role: pixel of the tile
  int G = gridDim.x;
  long long R = A.total_rounds;
  int n_list = 0;
  if constexpr (LIST) {
    // ACTIVE-TILE mode (sessd_conv2d_sk_active): the tile-space pixels are the 2x2 tiles of a device list (entry image * ntiles2 +
    // tile; 32 entries = the 128 pixels of a unit: lanes 0..63 the upper pixel rows of the 32 tiles, 64..127 the lower ones, so
    // that a run of adjacent tiles is a run of adjacent pixels), their number is on the device: the round list is sized HERE, in
    // shares of at least `min_rounds` rounds; the workgroups beyond that leave at once. The image is part of every thread's offsets.
    n_list = uni(min(A.n_list[0], A.list_cap));
    R = (long long)((n_list + 31) >> 5) * A.cgroups * A.rpg;
    if (R == 0) return;
    long long g = R / A.min_rounds;
    g = g < 1 ? 1 : (g > G ? G : g);
    G = (int)(g >= 8 ? (g & ~7LL) : g);
    if ((G & 7) ? ((int)blockIdx.x >= G) : ((int)(blockIdx.x >> 3) >= (G >> 3))) return;
  }
  // share w of the round list; consecutive shares on one XCD (workgroup b runs on XCD b % 8): neighbouring units share input
  // rows and weights through that XCD's L2
  const int w = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
  const int ra = (int)((long long)w * R / G), rb = (int)((long long)(w + 1) * R / G);
  if (ra >= rb) return;
  const int in_plane = A.hin * A.win;
  const unsigned plane4 = (unsigned)in_plane * 4u;
  const unsigned xbytes = (unsigned)A.cin * plane4;

  // ---- loader state: the chunk to fetch next, (class, batch element, pixel tile, cout group, channel block, tap)
  int lr = ra, lc = 0;
  int lnt, lcb, lt, lcg, lpt, lb;
  {
    const int g = uni(ra / A.rpg), rem = ra - g * A.rpg;
    for (int c = 1; c < A.nclass; ++c)
      if (rem >= Kp->cls[c].r_begin) lc = c;
    lc = uni(lc);
    lnt = uni(Kp->cls[lc].ntaps);
    const int rr = rem - Kp->cls[lc].r_begin;
    lcb = uni(rr / lnt);
    lt = rr - lcb * lnt;
    lcg = uni(g % A.cgroups);
    const int gb = uni(g / A.cgroups);
    lpt = LIST ? gb : uni(gb % A.ptiles);
    lb = LIST ? 0 : uni(gb / A.ptiles);
  }
  rsrc_t xr = LIST ? make_rsrc(A.in, (unsigned)A.batch * xbytes) : make_rsrc(A.in + (size_t)lb * A.cin * in_plane, xbytes);
  rsrc_t wr = make_rsrc(Kp->cls[lc].wpk, (unsigned)(A.cgroups * A.ncb * lnt) * (CSK_CHUNK_FLOATS * 4u));
  // this thread's LDS addresses: everything else is an immediate offset
  float* const st_w = lds + tid * 4;
  float* const st_x = lds + CSK_CHUNK_FLOATS + (hh * 256 + lp) * 4;
  const float* const rd_a = lds + (h * 256 + wc * 64 + j) * 4;
  const float* const rd_b = lds + CSK_CHUNK_FLOATS + (h * 256 + wp * 64 + j) * 4;
  float gx[2][8];   // two fetch register sets: a chunk is fetched two rounds before it is written to LDS
  f32x4v gw[2][2];
  unsigned voff;

#define SESSD_CSK_ENTER_UNIT()                                                                        \
  {                                                                                                   \
    const __attribute__((address_space(4))) SkArgs* Up = Kp;                                          \
    asm volatile("" : "+s"(Up));                                                                      \
    const int u_npix = Up->npix, u_wt = Up->wt, u_mul = Up->in_mul, u_hin = Up->hin, u_win = Up->win; \
    const int p_ = lpt * 128 + lp;                                                                    \
    bool live_ = p_ < u_npix;                                                                         \
    int y_ = live_ ? p_ / u_wt : 0, x_ = live_ ? p_ - (p_ / u_wt) * u_wt : 0;                         \
    unsigned img_ = 0u;                                                                               \
    if constexpr (LIST) {                                                                             \
      const int k_ = lpt * 32 + ((lp & 63) >> 1);                                                     \
      const int e_ = k_ < n_list ? Up->tile_list[k_] : -1;                                            \
      live_ = e_ >= 0;                                                                                \
      const int bl_ = live_ ? e_ / Up->ntiles2 : 0, t_ = live_ ? e_ - bl_ * Up->ntiles2 : 0;          \
      y_ = 2 * (t_ / Up->tw2) + (lp >> 6);                                                            \
      x_ = 2 * (t_ - (t_ / Up->tw2) * Up->tw2) + (lp & 1);                                            \
      img_ = (unsigned)bl_ * xbytes;                                                                  \
    }                                                                                                 \
    const int y0_ = y_ * u_mul, x0_ = x_ * u_mul;                                                     \
    for (int t_ = 0; t_ < lnt; ++t_) {                                                                \
      const int iy_ = y0_ + Up->cls[lc].dy[t_], ix_ = x0_ + Up->cls[lc].dx[t_];                       \
      const bool ok_ = live_ && iy_ >= 0 && iy_ < u_hin && ix_ >= 0 && ix_ < u_win;                   \
      xo_tab[t_ * 256 + tid] = ok_ ? img_ + (unsigned)((iy_ * u_win + ix_) * 4) : SESSD_OOB;          \
    }                                                                                                 \
  }
  // fetch the loader's chunk into register set GS (voff = this thread's input offset for the chunk's tap, read from xo_tab
  // one round earlier), in two pieces; then step the loader (ADVANCE) and read the next chunk's offset (NEXT_VOFF)
#define SESSD_CSK_ISSUE_X(GS)                                                                         \
  {                                                                                                   \
    const unsigned xs_ = (unsigned)(lcb * 16 + hh * 8) * plane4;                                      \
    _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_)                                                  \
      gx[GS][c_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)voff, (int)(xs_ + (unsigned)c_ * plane4), 0)); \
  }
#define SESSD_CSK_ISSUE_W(GS)                                                                         \
  {                                                                                                   \
    const unsigned ws_ = (unsigned)((lcg * A.ncb + lcb) * lnt + lt) * (CSK_CHUNK_FLOATS * 4u);        \
    gw[GS][0] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, tid * 16, (int)ws_, 0));          \
    gw[GS][1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, tid * 16 + 4096, (int)ws_, 0));   \
  }
#define SESSD_CSK_NEXT_VOFF() voff = xo_tab[lt * 256 + tid];
#define SESSD_CSK_ISSUE(GS)                                                                           \
  {                                                                                                   \
    SESSD_CSK_ISSUE_X(GS)                                                                             \
    SESSD_CSK_ISSUE_W(GS)                                                                             \
    SESSD_CSK_ADVANCE()                                                                               \
    SESSD_CSK_NEXT_VOFF()                                                                             \
  }
  // next chunk of the share (the last one is fetched again instead of running past the share's end)
#define SESSD_CSK_ADVANCE()                                                                           \
  if (lr + 1 < rb) {                                                                                  \
    ++lr;                                                                                             \
    if (++lt == lnt) {                                                                                \
      lt = 0;                                                                                         \
      if (++lcb == A.ncb) {                                                                           \
        lcb = 0;                                                                                      \
        const __attribute__((address_space(4))) SkArgs* Vp = Kp;                                      \
        asm volatile("" : "+s"(Vp));                                                                  \
        if (++lc == Vp->nclass) {                                                                     \
          lc = 0;                                                                                     \
          if (++lcg == Vp->cgroups) {                                                                 \
            lcg = 0;                                                                                  \
            ++lpt;                                                                                    \
            if constexpr (!LIST) {                                                                    \
              if (lpt == Vp->ptiles) {                                                                \
                lpt = 0;                                                                              \
                ++lb;                                                                                 \
                xr = make_rsrc(Vp->in + (size_t)lb * Vp->cin * in_plane, xbytes);                     \
              }                                                                                       \
            }                                                                                         \
          }                                                                                           \
        }                                                                                             \
        lc = uni(lc);                                                                                 \
        lnt = uni(Vp->cls[lc].ntaps);                                                                 \
        wr = make_rsrc(Vp->cls[lc].wpk, (unsigned)(Vp->cgroups * Vp->ncb * lnt) * (CSK_CHUNK_FLOATS * 4u)); \
        SESSD_CSK_ENTER_UNIT()                                                                        \
      }                                                                                               \
    }                                                                                                 \
  }
#define SESSD_CSK_STORE(GS, PB)                                                                       \
  {                                                                                                   \
    *reinterpret_cast<f32x4v*>(st_w + (PB) * (2 * CSK_CHUNK_FLOATS)) = gw[GS][0];                     \
    *reinterpret_cast<f32x4v*>(st_w + (PB) * (2 * CSK_CHUNK_FLOATS) + 1024) = gw[GS][1];              \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                \
      f32x4v v_;                                                                                      \
      v_.x = gx[GS][q_ * 4 + 0]; v_.y = gx[GS][q_ * 4 + 1]; v_.z = gx[GS][q_ * 4 + 2]; v_.w = gx[GS][q_ * 4 + 3]; \
      *reinterpret_cast<f32x4v*>(st_x + (PB) * (2 * CSK_CHUNK_FLOATS) + q_ * 512) = v_;               \
    }                                                                                                 \
  }
#define SESSD_CSK_READF(SET, PB)                                                                      \
  {                                                                                                   \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                                  \
      _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                              \
        fa[SET][t_][q_] = *reinterpret_cast<const f32x4v*>(rd_a + (PB) * (2 * CSK_CHUNK_FLOATS) + q_ * 512 + t_ * 128); \
        fb[SET][t_][q_] = *reinterpret_cast<const f32x4v*>(rd_b + (PB) * (2 * CSK_CHUNK_FLOATS) + q_ * 512 + t_ * 128); \
      }                                                                                               \
  }
// the 4 MFMAs of MFMA step S (0..7): channels S and 8 + S of the chunk
#define SESSD_CSK_MMA(SET, S)                                                                         \
  {                                                                                                   \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                                  \
      _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_)                                                \
        acc[c_][p_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][c_][(S) >> 2][(S) & 3], fb[SET][p_][(S) >> 2][(S) & 3], acc[c_][p_], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  // one round r with sets (SETC current, SETN next). A workgroup is one wave per SIMD and a wave issues one instruction per 4
  // cycles, so the ~110 non-MFMA instructions of a round are dealt out in pieces of <= 16 behind each group of four MFMAs (256
  // cycles of matrix-pipe time): issued as one block they leave the pipe idle (measured: 76 % MFMA-busy in the loop).
  //   fragments of round r+1 <- LDS; chunk r+3 fetch -> registers SETN; loader step; chunk r+2 (registers SETC, fetched during
  //   the previous round) -> the LDS buffer round r has left, half a round before the barrier asks for it
#define SESSD_CSK_ITER(SETC, SETN)                                                                    \
  {                                                                                                   \
    SESSD_CSK_MMA(SETC, 0)                                                                            \
    SESSD_CSK_READF(SETN, SETN)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    SESSD_CSK_MMA(SETC, 1)                                                                            \
    SESSD_CSK_ISSUE_X(SETN)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    SESSD_CSK_MMA(SETC, 2)                                                                            \
    SESSD_CSK_ISSUE_W(SETN)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    SESSD_CSK_MMA(SETC, 3)                                                                            \
    SESSD_CSK_ADVANCE()                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    SESSD_CSK_MMA(SETC, 4)                                                                            \
    SESSD_CSK_NEXT_VOFF()                                                                             \
    SESSD_CSK_STORE(SETC, SETC)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    SESSD_CSK_MMA(SETC, 5)                                                                            \
    SESSD_CSK_MMA(SETC, 6)                                                                            \
    SESSD_CSK_MMA(SETC, 7)                                                                            \
    SESSD_LDS_BARRIER();                                                                              \
  }

  f32x4v fa[2][2][2], fb[2][2][2];  // [set][tile][quad]
  f32x16 acc[2][2];

  // ---- pipeline fill: chunk ra -> buffer 0 (+ fragment set 0), chunk ra+1 -> buffer 1, chunk ra+2 in flight in register set 0
  SESSD_CSK_ENTER_UNIT()
  voff = xo_tab[lt * 256 + tid];
  SESSD_CSK_ISSUE(0)
  SESSD_CSK_ISSUE(1)
  __builtin_amdgcn_sched_barrier(0);
  SESSD_CSK_STORE(0, 0)
  __builtin_amdgcn_sched_barrier(0);
  SESSD_CSK_ISSUE(0)
  SESSD_LDS_BARRIER();
  SESSD_CSK_READF(0, 0)
  SESSD_CSK_STORE(1, 1)
  SESSD_LDS_BARRIER();
  int par = 0;  // register set = LDS buffer of the current round (they alternate together)

  int r = ra;
  while (r < rb) {
    // ---- the segment [r0, r0 + n) of the unit (group g, class c) this share holds
    const __attribute__((address_space(4))) SkArgs* Ep = Kp;
    asm volatile("" : "+s"(Ep));
    const int g = uni(r / Ep->rpg), rem = uni(r - g * Ep->rpg);
    int c = 0;
    for (int k = 1; k < Ep->nclass; ++k)
      if (rem >= Ep->cls[k].r_begin) c = k;
    c = uni(c);
    const int rpu = uni(Ep->cls[c].rpu);
    const int r0 = uni(rem - Ep->cls[c].r_begin);
    const int n = uni(min(rpu - r0, rb - r));
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    int i = 0;
    if (par == 0) {
      for (; i + 2 <= n; i += 2) {
        SESSD_CSK_ITER(0, 1)
        SESSD_CSK_ITER(1, 0)
      }
      if (i < n) {
        SESSD_CSK_ITER(0, 1)
        par = 1;
      }
    } else {
      for (; i + 2 <= n; i += 2) {
        SESSD_CSK_ITER(1, 0)
        SESSD_CSK_ITER(0, 1)
      }
      if (i < n) {
        SESSD_CSK_ITER(1, 0)
        par = 0;
      }
    }
    r += n;
    const __attribute__((address_space(4))) SkArgs* Fp = Kp;
    asm volatile("" : "+s"(Fp));

    // ---- the segment's result: a whole unit is finished here, a part goes to the scratch slot
    const int cg = uni(g % Fp->cgroups), gb = uni(g / Fp->cgroups);
    const int pt = LIST ? gb : uni(gb % Fp->ptiles), b = LIST ? 0 : uni(gb / Fp->ptiles);
    bool fin = (r0 == 0 && n == rpu);
    if (!fin) {
      const long long S = (long long)g * Fp->rpg + Fp->cls[c].r_begin;  // the unit's first round
      const int w_first = (int)(((S + 1) * G - 1) / R), w_last = (int)(((S + rpu) * G - 1) / R);
      const rsrc_t sr = make_rsrc(Fp->scratch, (unsigned)(2 * G) * (unsigned)CSK_SLOT_BYTES);
      const unsigned my_slot = (unsigned)(2 * w + (r0 == 0 ? 1 : 0)) * (unsigned)CSK_SLOT_BYTES;
      const unsigned lane_off = (unsigned)(wave * 16 * 64 + lane) * 16u;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4v v;
            v.x = acc[a][q][4 * g]; v.y = acc[a][q][4 * g + 1]; v.z = acc[a][q][4 * g + 2]; v.w = acc[a][q][4 * g + 3];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4g, v), sr,
                                                   (int)(my_slot + lane_off + (unsigned)(((a * 2 + q) * 4 + g) * 64 * 16)), 0,
                                                   SESSD_SYSTEM_SCOPE);
          }
      __builtin_amdgcn_s_waitcnt(0);  // every thread's part is acknowledged by memory
      __syncthreads();
      if (tid == 0) {
        unsigned* cnt = Fp->counters + g * Fp->nclass + c;
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == (unsigned)(w_last - w_first)) ? 1 : 0;
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag = last;
      }
      __syncthreads();
      fin = s_flag != 0;
      if (fin) {
        // the part counted last finishes the unit: all parts (its own included) in share order
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][q][e] = 0.f;
        for (int wq = w_first; wq <= w_last; ++wq) {
          const unsigned slot = (unsigned)(2 * wq + (wq == w_first ? 1 : 0)) * (unsigned)CSK_SLOT_BYTES;
#pragma unroll
          for (int a = 0; a < 2; ++a) {  // half a tile at a time: 32 registers of loads in flight (two workgroups share a CU's registers)
            f32x4v p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              p[e] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                                                    sr, (int)(slot + lane_off + (unsigned)((a * 8 + e) * 64 * 16)), 0, SESSD_SYSTEM_SCOPE));
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const f32x4v v = p[q * 4 + g];
                acc[a][q][4 * g] += v.x; acc[a][q][4 * g + 1] += v.y; acc[a][q][4 * g + 2] += v.z; acc[a][q][4 * g + 3] += v.w;
              }
          }
        }
      }
    }
    if (fin) {
      // D layout (32x32): column = lane & 31 (pixel), row = (e & 3) + 8 (e >> 2) + 4 h (cout). Branch-free: everything goes
      // through buffer resources, an element outside the image / beyond cout gets an out-of-range offset (loads return 0, stores
      // are dropped), so that all loads of a pass are in flight together and no store waits for a load of the next element.
      const int e_cout = Fp->cout, e_relu = Fp->relu, e_wt = Fp->wt, e_npix = Fp->npix, e_mul = Fp->out_mul;
      const int e_py = Fp->cls[c].py, e_px = Fp->cls[c].px, e_wout = Fp->wout;
      const unsigned oplane = (unsigned)(Fp->hout * Fp->wout);
      const float* e_scale = Fp->scale;
      const float* e_shift = Fp->shift;
      const float* e_res = Fp->residual;
      const size_t boff = (size_t)b * e_cout * oplane;
      const unsigned obytes = (unsigned)e_cout * oplane * 4u * (LIST ? (unsigned)Fp->batch : 1u);
      const rsrc_t orr = make_rsrc(Fp->out + boff, obytes);
      const rsrc_t rr = make_rsrc(e_res ? e_res + boff : Fp->out, e_res ? obytes : 0u);
      const rsrc_t scr = make_rsrc(e_scale ? e_scale : Fp->out, e_scale ? (unsigned)e_cout * 4u : 0u);
      const rsrc_t shr = make_rsrc(e_shift ? e_shift : Fp->out, e_shift ? (unsigned)e_cout * 4u : 0u);
      const int co0 = cg * 128 + wc * 64 + 4 * h;  // + a * 32 + (e & 3) + 8 * (e >> 2)
      // per-element offsets are built in VGPRs (an SGPR offset per element would need 64 live scalars); a quarter tile
      // (16 couts x 32 pixels per wave) at a time: two workgroups share a CU's registers
      const unsigned oplane4 = oplane * 4u;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int p = pt * 128 + wp * 64 + q * 32 + j;
        bool live = p < e_npix;
        int y = live ? p / e_wt : 0, x = live ? p - (p / e_wt) * e_wt : 0;
        unsigned oimg = 0u;
        if constexpr (LIST) {
          const int k = pt * 32 + ((q * 32 + j) >> 1);
          const int en = k < n_list ? Fp->tile_list[k] : -1;
          live = en >= 0;
          const int bl = live ? en / Fp->ntiles2 : 0, t = live ? en - bl * Fp->ntiles2 : 0;
          y = 2 * (t / Fp->tw2) + wp;
          x = 2 * (t - (t / Fp->tw2) * Fp->tw2) + (j & 1);
          oimg = (unsigned)bl * (unsigned)e_cout * oplane4;
        }
        const unsigned vbase = oimg + (unsigned)co0 * oplane4 + (unsigned)((y * e_mul + e_py) * e_wout + (x * e_mul + e_px)) * 4u;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          float scv[16], shv[16], rv[16];
          unsigned vo[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int k = a * 32 + (e & 3) + 8 * (e >> 2);
            const unsigned so = (unsigned)(co0 + k) * 4u;
            scv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scr, (int)so, 0, 0));
            shv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shr, (int)so, 0, 0));
            vo[e] = (live && co0 + k < e_cout) ? vbase + (unsigned)k * oplane4 : SESSD_OOB;
            rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)vo[e], 0, 0));
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = fmaf(acc[a][q][e], e_scale ? scv[e] : 1.f, shv[e]);
            if (e_relu) v = fmaxf(v, 0.f);
            if (e_res) v += rv[e];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, (int)vo[e], 0, 0);
          }
        }
      }
    }
  }
#undef SESSD_CSK_ENTER_UNIT
#undef SESSD_CSK_ISSUE
#undef SESSD_CSK_ISSUE_X
#undef SESSD_CSK_ISSUE_W
#undef SESSD_CSK_NEXT_VOFF
#undef SESSD_CSK_ADVANCE
#undef SESSD_CSK_STORE
#undef SESSD_CSK_READF
#undef SESSD_CSK_MMA
#undef SESSD_CSK_ITER
}

// out[(((cg * ncb + cb) * nt + t) * 2048) + ((h * 2 + q) * 128 + i) * 4 + e] = w[(cg*128 + i) * so + (cb*16 + 8h + 4q + e) * sc + off[t]]
struct SkPackArgs {
  const float* w;
  long long so, sc;
  int co, ci, nt;
  int off[16];
};
__global__ __launch_bounds__(256) void conv2d_sk_pack_kernel(SkPackArgs A, float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int ncb = A.ci / 16;
  const size_t total = (size_t)sessd_divup(A.co, 128) * ncb * A.nt * CSK_CHUNK_FLOATS;
  if (idx >= total) return;
  const int e = (int)(idx & 3), i = (int)((idx >> 2) & 127), hq = (int)((idx >> 9) & 3);
  size_t rest = idx >> 11;
  const int t = (int)(rest % A.nt);
  rest /= A.nt;
  const int cb = (int)(rest % ncb), cg = (int)(rest / ncb);
  const int o = cg * 128 + i, ch = cb * 16 + (hq >> 1) * 8 + (hq & 1) * 4 + e;
  out[idx] = o < A.co ? A.w[(long long)o * A.so + (long long)ch * A.sc + A.off[t]] : 0.f;
}

int default_workgroups(int* out) {
  int dev = 0, cus = 0;
  SESSD_TRY(hipGetDevice(&dev));
  SESSD_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  *out = cus & ~7;
  return *out >= 8 ? SESSD_OK : SESSD_EINVAL;
}

int launch_conv2d_sk(const float* in, int batch, int cin, int hin, int win, int nclass, const float* const* wpk,
                     const int* ntaps, const int* taps_dy, const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out,
                     int cout, int hout, int wout, int out_mul, const int* out_py, const int* out_px, const float* scale,
                     const float* shift, int relu, const float* residual, void* workspace, size_t workspace_bytes, int workgroups,
                     hipStream_t stream, const int* tile_list, const int* n_list, int list_cap, int min_rounds) {
  if (batch < 1 || cin < 16 || cin % 16 || cout < 1 || nclass < 1 || nclass > 4 || workgroups < 0 || (workgroups & 7)) return SESSD_EINVAL;
  // a batch element's input and output are addressed through 32-bit buffer offsets
  if ((long long)cin * hin * win * 4 >= 0x7fffffffLL || (long long)cout * hout * wout * 4 >= 0x7fffffffLL) return SESSD_EINVAL;
  // active-tile mode: the image is part of the offsets too, and the tile space is cut into 2x2 tiles
  if (tile_list && (!n_list || list_cap < 1 || (tile_h & 1) || (tile_w & 1) || (long long)batch * cin * hin * win * 4 >= 0x7fffffffLL ||
                    (long long)batch * cout * hout * wout * 4 >= 0x7fffffffLL))
    return SESSD_EINVAL;
  if (workgroups == 0) {
    const int rc = default_workgroups(&workgroups);
    if (rc != SESSD_OK) return rc;
  }
  SkArgs A;
  A.in = in; A.out = out; A.scale = scale; A.shift = shift; A.residual = residual;
  A.cin = cin; A.hin = hin; A.win = win; A.cout = cout; A.hout = hout; A.wout = wout; A.wt = tile_w;
  A.in_mul = in_mul; A.out_mul = out_mul; A.relu = relu;
  A.npix = tile_h * tile_w; A.ptiles = sessd_divup(A.npix, 128); A.cgroups = sessd_divup(cout, 128); A.ncb = cin / 16;
  A.batch = batch; A.nclass = nclass;
  A.tile_list = tile_list; A.n_list = n_list; A.list_cap = list_cap; A.min_rounds = min_rounds < 1 ? 1 : min_rounds;
  A.ntiles2 = (tile_h / 2) * (tile_w / 2); A.tw2 = tile_w / 2;
  // the round list: for every (batch element, pixel tile, cout group) the units of all classes one after the other (more taps
  // first) -- the classes of a pixel tile read the same input and write the interleaved pixels of the same output lines, and
  // consecutive shares run on the same XCD: the input is fetched once and the half-written output lines meet in that L2
  int order[4] = {0, 1, 2, 3};
  for (int a = 0; a < nclass; ++a)
    for (int b = a + 1; b < nclass; ++b)
      if (ntaps[order[b]] > ntaps[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
  const long long groups = (long long)batch * A.ptiles * A.cgroups;  // (batch element, pixel tile, cout group)
  int rpg = 0;
  for (int k = 0; k < nclass; ++k) {
    const int c = order[k];
    if (ntaps[c] < 1 || ntaps[c] > 9) return SESSD_EINVAL;
    SkClass& C = A.cls[k];
    C.wpk = wpk[c]; C.ntaps = ntaps[c]; C.py = out_py[c]; C.px = out_px[c];
    C.rpu = ntaps[c] * A.ncb;
    C.r_begin = rpg; C.u_begin = 0;
    for (int t = 0; t < 9; ++t) {
      C.dy[t] = t < ntaps[c] ? taps_dy[9 * c + t] : 0;
      C.dx[t] = t < ntaps[c] ? taps_dx[9 * c + t] : 0;
    }
    rpg += C.rpu;
  }
  for (int k = nclass; k < 4; ++k) A.cls[k] = A.cls[0];
  A.rpg = rpg;
  const long long rounds = groups * rpg;
  if (rounds > 0x7fffffffLL || groups * nclass > 0x7fffffffLL) return SESSD_EINVAL;
  A.total_rounds = (int)rounds;
  const size_t cbytes = sessd_align((size_t)groups * nclass * 4, 256);
  if (cbytes + (size_t)2 * workgroups * CSK_SLOT_BYTES > workspace_bytes) return SESSD_EWORKSPACE;
  // every share must hold at least one round (the part count of a cut unit is a difference of share indices)
  if (workgroups > A.total_rounds) workgroups = A.total_rounds >= 8 ? (A.total_rounds & ~7) : A.total_rounds;
  A.counters = (unsigned*)workspace;
  A.scratch = (float*)((char*)workspace + cbytes);
  if (tile_list)   // (shares are sized on the device: the workspace and the launch are the dense layer's)
    SESSD_LAUNCH((conv2d_sk_kernel<true>), dim3(workgroups), dim3(256), 0, stream, A);
  else
    SESSD_LAUNCH((conv2d_sk_kernel<false>), dim3(workgroups), dim3(256), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // namespace

extern "C" {

// Scratch + counters of sessd_conv2d_sk for a launch over `nclass` classes of (batch, tile_h x tile_w pixels, cout).
// workgroups 0 = one per CU. The caller zeroes the workspace ONCE (the kernel leaves the counters zero) and must not share it
// between launches that may run concurrently.
size_t sessd_conv2d_sk_workspace_bytes(int batch, int tile_h, int tile_w, int cout, int nclass, int workgroups) {
  if (batch < 1 || tile_h < 1 || tile_w < 1 || cout < 1 || nclass < 1 || nclass > 4 || workgroups < 0) return 0;
  if (workgroups == 0 && default_workgroups(&workgroups) != SESSD_OK) return 0;
  const size_t units = (size_t)nclass * batch * sessd_divup(tile_h * tile_w, 128) * sessd_divup(cout, 128);
  return sessd_align(units * 4, 256) + (size_t)2 * workgroups * CSK_SLOT_BYTES;
}

// Weight of one class in the kernel's LDS image order: out [ceil(cout/128)][cin/16][ntaps][2048] with
// element ((h*2+q)*128 + i)*4 + e of a chunk = w[(cg*128 + i) * out_stride + (cb*16 + 8h + 4q + e) * in_stride + tap_offsets[t]]
// (element strides / offsets into w: a transposed, flipped or tap-selected view needs no intermediate); cin % 16 == 0, ntaps <= 16.
int sessd_conv2d_sk_pack(const float* w, long long out_stride, long long in_stride, const int* tap_offsets, int ntaps, int cout,
                         int cin, float* out, hipStream_t stream) {
  if (ntaps < 1 || ntaps > 16 || cout < 1 || cin < 16 || cin % 16) return SESSD_EINVAL;
  SkPackArgs P;
  P.w = w; P.so = out_stride; P.sc = in_stride; P.co = cout; P.ci = cin; P.nt = ntaps;
  for (int t = 0; t < 16; ++t) P.off[t] = t < ntaps ? tap_offsets[t] : 0;
  const size_t total = (size_t)sessd_divup(cout, 128) * (cin / 16) * ntaps * CSK_CHUNK_FLOATS;
  SESSD_LAUNCH(conv2d_sk_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, P, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// nclass (1..4) convolutions that share input, shapes and epilogue but not weights / taps / output phase, in ONE stream-K
// launch: a Conv2d (nclass 1: 3x3 or 1x1, any stride via in_mul) or the four output-parity classes of the stride-2 transposed
// conv. Class c: wpk[c] (device, sessd_conv2d_sk_pack), ntaps[c] <= 9 taps with input offsets taps_dy/dx[9 c + t] (host ints),
// output pixel (y * out_mul + out_py[c], x * out_mul + out_px[c]) for tile-space pixel (y, x) of tile_h x tile_w, input pixel
// (y * in_mul + dy, x * in_mul + dx). cin % 16 == 0. workgroups: a multiple of 8, 0 = one per CU.
int sessd_conv2d_sk(const float* in, int batch, int cin, int hin, int win, int nclass, const float* const* wpk,
                    const int* ntaps, const int* taps_dy, const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out,
                    int cout, int hout, int wout, int out_mul, const int* out_py, const int* out_px, const float* scale,
                    const float* shift, int relu, const float* residual, void* workspace, size_t workspace_bytes, int workgroups,
                    hipStream_t stream) {
  return launch_conv2d_sk(in, batch, cin, hin, win, nclass, wpk, ntaps, taps_dy, taps_dx, in_mul, tile_h, tile_w, out, cout, hout, wout,
                          out_mul, out_py, out_px, scale, shift, relu, residual, workspace, workspace_bytes, workgroups, stream, nullptr,
                          nullptr, 0, 1);
}

// The same launch over the listed 2x2 tiles of the TILE SPACE only (active-tile mode, csrc/dense_active.hip: the BEV maps of the
// SSFA neck are a per-channel constant away from the sparse sites): tile_list[0 .. min(*n_list, list_cap)) entries image *
// (tile_h/2 * tile_w/2) + tile in any order, count on the device; every class is computed at the four tile-space pixels of a
// listed tile, the other output pixels are left alone (sessd_fill_inactive_tiles writes the layer's constant there). Even tile_h,
// tile_w. Shares of the round list are at least min_rounds rounds long; the workgroups beyond rounds / min_rounds leave at once.
// Same packed weights, same workspace (sized for the dense layer) as sessd_conv2d_sk.
int sessd_conv2d_sk_active(const float* in, int batch, int cin, int hin, int win, int nclass, const float* const* wpk,
                           const int* ntaps, const int* taps_dy, const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out,
                           int cout, int hout, int wout, int out_mul, const int* out_py, const int* out_px, const float* scale,
                           const float* shift, int relu, const float* residual, const int32_t* tile_list, const int32_t* n_list,
                           int list_cap, int min_rounds, void* workspace, size_t workspace_bytes, int workgroups, hipStream_t stream) {
  if (!tile_list || !n_list || list_cap < 1) return SESSD_EINVAL;
  return launch_conv2d_sk(in, batch, cin, hin, win, nclass, wpk, ntaps, taps_dy, taps_dx, in_mul, tile_h, tile_w, out, cout, hout, wout,
                          out_mul, out_py, out_px, scale, shift, relu, residual, workspace, workspace_bytes, workgroups, stream, tile_list,
                          n_list, list_cap, min_rounds);
}

}  // extern "C"
