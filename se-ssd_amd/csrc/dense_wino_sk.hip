// 3x3 stride-1 BEV convolutions (the seven of the SSFA neck, det3d/models/necks/rpn_v1.py:135-210) as fused Winograd
// F(2x2,3x3) on the f32 matrix cores, decomposed "stream-K": second generation of conv3x3s1_winograd_kernel
// (dense_conv.hip), written after its counters (profiles/r2_wino_pmc.txt): f32 MFMA and VALU never co-execute on gfx950
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so the 3.4 VALU instructions per MFMA of the patch transform -- repeated by each of
// the four 32-cout workgroups of a tile block -- cost a quarter of the SIMD time, 1100 workgroups over 256 CUs leave 16 %
// of the CU time idle in the last wave of workgroups, and operands arrive at 0.5 KB-loads per MFMA.
//
//   unit      = 32 consecutive 2x2-output tiles x (CBN x 32) couts x all input channels; 16 GEMMs M_xi = U_xi V_xi
//   workgroup = NW waves; wave w owns the 16/NW transform points xi = XW w .. XW w + XW-1 for ALL couts of the unit
//               (XW x CBN = 8 accumulators of 32x32). Two shapes: <8 waves, 128 couts> (one workgroup per CU) and
//               <4 waves, 64 couts> (two per CU: one's epilogue and pipeline refill hide behind the other's MFMAs).
//   round     = NW k-steps (2 NW input channels): wave w loads the 4x4 patches of k-step w (lane = tile, channel parity;
//               4 x 16-B buffer loads, out-of-image rows by out-of-range offsets), transforms them in registers and writes
//               V[k-step w][16 xi] of the NEXT round into the other LDS buffer -- once per unit, not once per 32 couts;
//               its 8 NW MFMAs of THIS round take B (V) from LDS and A (U = G g G^T, host-packed so the 8 operands of a
//               lane and k-step are two 16-B loads) through a register ring of NW sets, loaded NW - 1 k-steps ahead with
//               exact vmcnt. One barrier per round.
//   stream-K  : the launch has a fixed number of persistent workgroups (a multiple of 8); the list of all rounds of all
//               units is cut into equal contiguous shares, so every CU runs the same number of rounds (275 units on 256 CUs
//               would otherwise take two passes). A unit cut by a share boundary is finished by whichever of its parts
//               arrives last: a part that finds all others already counted adds their partial outputs (prefetched while
//               it transforms its own) and finishes; otherwise it applies the (linear) output transform to its partial
//               sums, writes them to its scratch slot with system-scope write-through stores and bumps the unit's counter
//               -- if that made it the last after all, it re-reads all parts in share order. No workgroup ever waits for
//               another. Summation order is fixed (share order), so results do not depend on arrival order.
// Numerics: Winograd rounding (1e-6 of the output scale); the bits depend on (shape, number of workgroups), not on timing.
#include <cstdlib>
#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using f32x2v = __attribute__((ext_vector_type(2))) float;
typedef unsigned int u32x4g __attribute__((__vector_size__(16)));  // the b128 buffer builtins' own type

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
#define SESSD_OOB 0x80000000u
// Workgroup barrier that orders LDS traffic only. __syncthreads() is a workgroup-scope fence: with global stores in flight
// hipcc emits s_waitcnt vmcnt(0) in front of it, i.e. every barrier of the epilogue would wait for the previous pass's output
// stores to be acknowledged by memory (measured: 3 us per pass).
#define SESSD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SESSD_SYSTEM_SCOPE 17  // sc0 | sc1: write-through to / read from memory, past the per-XCD L2

// a float add the SLP vectoriser cannot pair into v_pk_add_f32 (variant bit 0 of the kernel below)
__device__ __forceinline__ float sadd(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct WinoArgs {
  const float* in;        // (B, cin, H, W)
  const float* upk;       // [cout groups][cin/2][wave NW][h 2][j 32][cb CBN][xi_local XW]
  float* out;             // (B, cout, H, W)
  const float* scale;
  const float* shift;
  const float* residual;
  float* scratch;         // [2 * workgroups][CBN*32 couts][32 tiles][4]
  unsigned* counters;     // [units], zero between launches
  int cin, hin, win, cout, relu;
  int tw, ntiles, tblocks, ngroups, rpu, total_rounds;
  int bper;               // batch elements per weight set (several layers of one shape in one launch: set = b / bper)
  int ss_stride;          // floats between the sets' scale / shift vectors (0 with one set)
  long long upk_stride;   // floats between the sets' packed U
  // active-tile mode (LIST kernels): entries image * ntiles + tile, their count on the device
  const int* tile_list;
  const int* n_list;
  int list_cap, min_rounds, batch;
};

// VAR: measured code variants of the main loop (same arithmetic, same results; `SESSD_WINO_VAR` selects one at launch for
// A/B runs, 0 ships): bit 0 = the patch transform with scalar adds instead of packed v_pk_add_f32 (MI355X_MICROARCH.md
// measures packed adds beside MFMAs as an anti-lever), bit 1 = s_setprio around the MFMA groups.
template <int NW, int CBN, int VAR = 0, bool LIST = false>
__global__ __launch_bounds__(NW * 64, 8 / NW) void conv3x3s1_winograd_sk_kernel(WinoArgs A) {
  constexpr int XW = 16 / NW;                          // transform points per wave
  static_assert(XW * CBN == 8, "8 accumulators per wave");
  constexpr int NT = NW * 64;
  constexpr unsigned WSTEP = NW * 2u * 32u * 32u;      // bytes of packed U per k-step
  constexpr int VBUF = NW * 1024;                      // floats of one V buffer: [ks NW][xi 16][h 2][tile 32]
  constexpr int SLOT = CBN * 32 * 32 * 4;              // floats of one scratch slot
  constexpr int NPT = 1024 / NT;                       // (cout, tile) pairs per thread and 32-cout pass
  constexpr int RING = NW;                             // U operand sets in flight (one round's worth)
  __shared__ __attribute__((aligned(16))) float lds[16384];  // 64 KB: V double buffer, then the M exchange of the epilogue
  __shared__ int s_last, s_pend;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  int G = gridDim.x;
  long long R = A.total_rounds;
  int n_list = 0;
  if constexpr (LIST) {
    // ACTIVE-TILE mode (sessd_conv3x3_winograd_sk_active): the tiles are the entries (image * ntiles + tile) of a device list,
    // their number is on the device, so the round list is sized HERE. A sparse frame leaves fewer rounds than the launch has
    // workgroups could usefully cut: shares of at least `min_rounds` rounds, the workgroups beyond that leave at once.
    n_list = __builtin_amdgcn_readfirstlane(min(A.n_list[0], A.list_cap));
    R = (long long)((n_list + 31) >> 5) * A.ngroups * A.rpu;
    if (R == 0) return;
    if (A.min_rounds < 0) {
      // WHOLE-UNIT shares (round 5; min_rounds = -k: at least k units per workgroup): no unit is ever cut, so no partial sum
      // crosses memory (the cut-unit exchange of short shares was 4 of the 5x algorithmic traffic of a list launch,
      // profiles/r4s2_wino_traffic.json) and a layer with few listed tiles occupies as many workgroups as it has units -- the
      // other CUs stay free for the other frame in flight. One workgroup per unit while the launch has enough of them.
      const long long U = R / A.rpu;
      long long g = U / (-A.min_rounds);
      g = g < 1 ? 1 : (g > G ? G : g);
      G = (int)g;
      if ((int)blockIdx.x >= G) return;
    } else {
      long long g = R / A.min_rounds;
      g = g < 1 ? 1 : (g > G ? G : g);
      G = (int)(g >= 8 ? (g & ~7LL) : g);
      if ((G & 7) ? ((int)blockIdx.x >= G) : ((int)(blockIdx.x >> 3) >= (G >> 3))) return;
    }
  }
  const bool whole_units = LIST && A.min_rounds < 0;
  // share w of the round list; consecutive shares on one XCD (workgroup b runs on XCD b % 8): neighbouring units share
  // input rows through that XCD's L2
  const int w = (whole_units || (G & 7)) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
  int r = whole_units ? (int)((long long)w * (R / A.rpu) / G) * A.rpu : (int)((long long)w * R / G);
  const int r_stop = whole_units ? (int)((long long)(w + 1) * (R / A.rpu) / G) * A.rpu : (int)((long long)(w + 1) * R / G);
  const int in_plane = A.hin * A.win;
  const size_t out_plane = (size_t)in_plane;
  const unsigned xstep = 2u * (unsigned)in_plane * 4u;  // bytes per k-step (2 channels)
  const unsigned wo = (unsigned)(((wave * 2 + h) * 32 + j) * 32);

  // A cut unit this workgroup has written a part of but not finished: 0 none, 1 part stored, 2 part counted and found to be
  // the last one (this workgroup finishes the unit after its last segment)
  int pend_state = 0, pend_u = 0, pend_first = 0, pend_last = 0, pend_tbase = 0, pend_mbase = 0, pend_b = 0;
  // Every vector-memory operation issued before this point has completed in every thread (vmcnt returns in order; called right
  // after a wait that covers younger loads, or with an explicit wait) -> count the stored part.
#define SESSD_SK_SIGNAL()                                                                          \
  {                                                                                                \
    __builtin_amdgcn_s_waitcnt(0);                                                                 \
    __syncthreads();                                                                               \
    if (tid == 0) {                                                                                \
      const unsigned old = __hip_atomic_fetch_add(A.counters + pend_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
      s_pend = (old == (unsigned)(pend_last - pend_first)) ? 1 : 0;                                \
    }                                                                                              \
    __syncthreads();                                                                               \
    pend_state = s_pend ? 2 : 0;                                                                   \
  }

#define SESSD_SK_FINALIZE(Y, CO, SC, SH)                                                            \
  if (tok && (CO) < e_cout) {                                                                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                 \
      const size_t o = (size_t)(CO) * out_plane + pix + (size_t)a * e_win;                          \
      float v0 = fmaf((Y)[2 * a], (SC), (SH)), v1 = fmaf((Y)[2 * a + 1], (SC), (SH));              \
      if (e_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }                                 \
      if (resb) { v0 += resb[o]; v1 += resb[o + 1]; }                                               \
      *reinterpret_cast<float2*>(outb + o) = make_float2(v0, v1);                                   \
    }                                                                                               \
  }
// Finish the cut unit this workgroup turned out to be the last part of: all parts (its own included) from the scratch slots,
// added in share order, then BatchNorm / ReLU / residual and the stores.
#define SESSD_SK_FINISH_PENDING()                                                                  \
  {                                                                                                \
    const __attribute__((address_space(4))) WinoArgs* Fp =                                         \
        (const __attribute__((address_space(4))) WinoArgs*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(Fp));                                                                   \
    const int f_set = pend_b / Fp->bper;                                                           \
    const float* f_scale = Fp->scale ? Fp->scale + (size_t)f_set * Fp->ss_stride : nullptr;        \
    const float* f_shift = Fp->shift ? Fp->shift + (size_t)f_set * Fp->ss_stride : nullptr;        \
    const int e_cout = Fp->cout, e_relu = Fp->relu, e_win = Fp->win, f_tw = Fp->tw;                \
    const rsrc_t fr = make_rsrc(Fp->scratch, (unsigned)(2 * G) * SLOT * 4u);                       \
    float* outb = Fp->out + (size_t)pend_b * e_cout * out_plane;                                   \
    const float* resb = Fp->residual ? Fp->residual + (size_t)pend_b * e_cout * out_plane : nullptr; \
    const int tl = tid & 31, col0 = tid >> 5;                                                      \
    int tt = pend_tbase + tl;                                                                      \
    bool tok = tt < Fp->ntiles;                                                                    \
    size_t pix = 0;                                                                                \
    if constexpr (LIST) {                                                                          \
      const int fe = tt < n_list ? Fp->tile_list[tt] : -1;                                         \
      tok = fe >= 0;                                                                               \
      const int fb = tok ? fe / Fp->ntiles : 0;                                                    \
      tt = tok ? fe - fb * Fp->ntiles : 0;                                                         \
      pix = (size_t)fb * e_cout * out_plane;                                                       \
    }                                                                                              \
    const int oty = tok ? tt / f_tw : 0, otx = tok ? tt - (tt / f_tw) * f_tw : 0;                  \
    pix += (size_t)(2 * oty) * e_win + 2 * otx;                                                    \
    _Pragma("unroll 1") for (int cb = 0; cb < CBN; ++cb) {                                         \
      f32x4v ysum[NPT];                                                                            \
      _Pragma("unroll") for (int n = 0; n < NPT; ++n) ysum[n] = 0.f;                               \
      for (int wq = pend_first; wq <= pend_last; ++wq) {                                           \
        const unsigned slot = (unsigned)(2 * wq + (wq == pend_first ? 1 : 0)) * (unsigned)(SLOT * 4); \
        f32x4v p[NPT];                                                                             \
        _Pragma("unroll") for (int n = 0; n < NPT; ++n)                                            \
          p[n] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(                 \
              fr, (int)(slot + (unsigned)(((cb * 32 + col0 + (NT / 32) * n) * 32 + tl) * 16)), 0, SESSD_SYSTEM_SCOPE)); \
        _Pragma("unroll") for (int n = 0; n < NPT; ++n) ysum[n] += p[n];                           \
      }                                                                                            \
      _Pragma("unroll") for (int n = 0; n < NPT; ++n) {                                            \
        const int co = pend_mbase + cb * 32 + col0 + (NT / 32) * n;                                \
        const float scv = (f_scale && co < e_cout) ? f_scale[co] : 1.f, shv = (f_shift && co < e_cout) ? f_shift[co] : 0.f; \
        const float z[4] = {ysum[n].x, ysum[n].y, ysum[n].z, ysum[n].w};                           \
        SESSD_SK_FINALIZE(z, co, scv, shv)                                                         \
      }                                                                                            \
    }                                                                                              \
    if (tid == 0) __hip_atomic_store(Fp->counters + pend_u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    pend_state = 0;                                                                                \
  }

  while (r < r_stop) {
    // everything that shapes a buffer resource or an SGPR offset is forced wave-uniform (otherwise hipcc wraps every
    // buffer load in a readfirstlane waterfall loop; integer division runs on the VALU)
    const int u = __builtin_amdgcn_readfirstlane(r / A.rpu);
    const int r0 = __builtin_amdgcn_readfirstlane(r - u * A.rpu);
    const int r1 = __builtin_amdgcn_readfirstlane(min(r_stop - u * A.rpu, A.rpu));
    r = u * A.rpu + r1;
    const int cg = __builtin_amdgcn_readfirstlane(u % A.ngroups), ub = u / A.ngroups;
    // LIST: the tile blocks run over the whole list (a block may span images: the image is part of every lane's offsets), one
    // weight set
    const int tb = LIST ? __builtin_amdgcn_readfirstlane(ub) : __builtin_amdgcn_readfirstlane(ub % A.tblocks);
    const int b = LIST ? 0 : __builtin_amdgcn_readfirstlane(ub / A.tblocks);
    const int t_base = tb * 32, m_base = cg * (CBN * 32);
    const rsrc_t xr = LIST ? make_rsrc(A.in, (unsigned)A.batch * (unsigned)A.cin * (unsigned)in_plane * 4u)
                           : make_rsrc(A.in + (size_t)b * A.cin * in_plane, (unsigned)A.cin * in_plane * 4u);
    const int wset = LIST ? 0 : __builtin_amdgcn_readfirstlane(b / A.bper);
    const rsrc_t wr = make_rsrc(A.upk + (size_t)wset * A.upk_stride + (size_t)cg * (A.cin >> 1) * (WSTEP / 4), (unsigned)(A.cin >> 1) * WSTEP);

    // ---- transform role: lane = (tile j, channel parity h)
    int t = t_base + j;
    bool tlive = t < A.ntiles;
    unsigned img_off = 0;   // LIST: byte offset of the lane's image
    if constexpr (LIST) {
      const int e = t < n_list ? A.tile_list[t] : -1;
      tlive = e >= 0;
      const int bl = tlive ? e / A.ntiles : 0;
      t = tlive ? e - bl * A.ntiles : 0;
      img_off = (unsigned)bl * (unsigned)A.cin * (unsigned)in_plane * 4u;
    }
    const int ty = tlive ? t / A.tw : 0, tx = tlive ? t - (t / A.tw) * A.tw : 0;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    unsigned ro[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = y0 + q;
      ro[q] = (tlive && y >= 0 && y < A.hin) ? img_off + (unsigned)((h * in_plane + y * A.win + max(x0, 0)) * 4) : SESSD_OOB;
    }
    const bool mask_l = (tx == 0), mask_r = (tx == A.tw - 1);
    const bool edge = __builtin_amdgcn_ballot_w64(tlive && (mask_l || mask_r)) != 0;
    const int klast = r1 * NW - 1;

    f32x16 acc[XW][CBN];
#pragma unroll
    for (int x = 0; x < XW; ++x)
#pragma unroll
      for (int c = 0; c < CBN; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[x][c][q] = 0.f;
    f32x4v pr[4];
    f32x4v ua[RING][2];
    float bv[2][XW];

#define SESSD_SK_LOADP(ROUND)                                                                      \
  {                                                                                                \
    const unsigned xs = (unsigned)(min((ROUND), r1 - 1) * NW + wave) * xstep;                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
      pr[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)ro[q], (int)xs, 0)); \
  }
#define SESSD_SK_LOADU(SET, KG)                                                                    \
  {                                                                                                \
    const unsigned ws = (unsigned)min((KG), klast) * WSTEP;                                        \
    ua[SET][0] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wo, (int)ws, 0));        \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    ua[SET][1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(wo + 16u), (int)ws, 0)); \
  }
#define SESSD_SK_TRANSFORM(VOFF)                                                                   \
  {                                                                                                \
    if (edge) {                                                                                    \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
        const f32x4v p = pr[q];                                                                    \
        pr[q].x = mask_l ? 0.f : p.x; pr[q].y = mask_l ? p.x : p.y;                                \
        pr[q].z = mask_l ? p.y : p.z; pr[q].w = mask_l ? p.z : (mask_r ? 0.f : p.w);               \
      }                                                                                            \
    }                                                                                              \
    f32x2v tl[4], tr[4];                                                                           \
    if constexpr (VAR & 1) {                                                                       \
      tl[0].x = sadd(pr[0].x, -pr[2].x); tl[0].y = sadd(pr[0].y, -pr[2].y); tr[0].x = sadd(pr[0].z, -pr[2].z); tr[0].y = sadd(pr[0].w, -pr[2].w); \
      tl[1].x = sadd(pr[1].x, pr[2].x); tl[1].y = sadd(pr[1].y, pr[2].y); tr[1].x = sadd(pr[1].z, pr[2].z); tr[1].y = sadd(pr[1].w, pr[2].w); \
      tl[2].x = sadd(pr[2].x, -pr[1].x); tl[2].y = sadd(pr[2].y, -pr[1].y); tr[2].x = sadd(pr[2].z, -pr[1].z); tr[2].y = sadd(pr[2].w, -pr[1].w); \
      tl[3].x = sadd(pr[1].x, -pr[3].x); tl[3].y = sadd(pr[1].y, -pr[3].y); tr[3].x = sadd(pr[1].z, -pr[3].z); tr[3].y = sadd(pr[1].w, -pr[3].w); \
    } else {                                                                                       \
    tl[0] = pr[0].xy - pr[2].xy; tr[0] = pr[0].zw - pr[2].zw;                                      \
    tl[1] = pr[1].xy + pr[2].xy; tr[1] = pr[1].zw + pr[2].zw;                                      \
    tl[2] = pr[2].xy - pr[1].xy; tr[2] = pr[2].zw - pr[1].zw;                                      \
    tl[3] = pr[1].xy - pr[3].xy; tr[3] = pr[1].zw - pr[3].zw;                                      \
    }                                                                                              \
    float* dst = &lds[(VOFF) + wave * 1024 + h * 32 + j];                                          \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                \
      dst[(a * 4 + 0) * 64] = tl[a].x - tr[a].x;                                                   \
      dst[(a * 4 + 1) * 64] = tl[a].y + tr[a].x;                                                   \
      dst[(a * 4 + 2) * 64] = tr[a].x - tl[a].y;                                                   \
      dst[(a * 4 + 3) * 64] = tl[a].y - tr[a].y;                                                   \
    }                                                                                              \
  }
#define SESSD_SK_READV(P, KS, VOFF)                                                                \
  {                                                                                                \
    const float* vb = &lds[(VOFF) + (KS)*1024 + wave * (XW * 64) + h * 32 + j];                    \
    _Pragma("unroll") for (int x = 0; x < XW; ++x) bv[P][x] = vb[x * 64];                          \
  }
#define SESSD_SK_MMA(SET, P)                                                                       \
  {                                                                                                \
    if constexpr (VAR & 2) __builtin_amdgcn_s_setprio(3);                                          \
    _Pragma("unroll") for (int c = 0; c < CBN; ++c)                                                \
      _Pragma("unroll") for (int x = 0; x < XW; ++x)                                               \
        acc[x][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[SET][(c * XW + x) >> 2][(c * XW + x) & 3], bv[P][x], acc[x][c], 0, 0, 0); \
    if constexpr (VAR & 2) __builtin_amdgcn_s_setprio(0);                                          \
  }
  // step KS of a round: k-step kg0 + KS from ring set KS; the set freed by the previous step receives k-step + RING - 1
#define SESSD_SK_STEP(KS)                                                                          \
  {                                                                                                \
    SESSD_SK_LOADU(((KS) + RING - 1) & (RING - 1), kg0 + (KS) + RING - 1)                          \
    if ((KS) < NW - 1) SESSD_SK_READV(((KS) + 1) & 1, (KS) + 1, voff)                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_SK_MMA((KS) & (RING - 1), (KS) & 1)                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
#define SESSD_SK_ROUND()                                                                           \
  {                                                                                                \
    const int kg0 = rr * NW;                                                                       \
    SESSD_SK_LOADP(rr + 1)                                                                         \
    SESSD_SK_READV(0, 0, voff)                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_SK_STEP(0)                                                                               \
    SESSD_SK_STEP(1)                                                                               \
    SESSD_SK_STEP(2)                                                                               \
    SESSD_SK_STEP(3)                                                                               \
    if constexpr (NW == 8) {                                                                       \
      SESSD_SK_STEP(4)                                                                             \
      SESSD_SK_STEP(5)                                                                             \
      SESSD_SK_STEP(6)                                                                             \
      SESSD_SK_STEP(7)                                                                             \
    }                                                                                              \
    SESSD_SK_TRANSFORM(voff ^ VBUF)                                                                \
    __syncthreads();                                                                               \
    voff ^= VBUF;                                                                                  \
    ++rr;                                                                                          \
  }

    SESSD_SK_LOADP(r0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) {  // in ring order: the in-loop vmcnt ladder assumes it
      SESSD_SK_LOADU(s, r0 * NW + s)
      __builtin_amdgcn_sched_barrier(0);
    }
    SESSD_SK_TRANSFORM(0)
    __syncthreads();
    if (pend_state == 1) {
      // the previous segment's part: its stores are older than the patch loads every thread has just consumed
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(A.counters + pend_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_pend = (old == (unsigned)(pend_last - pend_first)) ? 1 : 0;
      }
      pend_state = 3;  // counted; the answer is read at this segment's epilogue
    }
    int voff = 0;
    int rr = r0;
    while (rr < r1) SESSD_SK_ROUND()
#undef SESSD_SK_LOADP
#undef SESSD_SK_LOADU
#undef SESSD_SK_TRANSFORM
#undef SESSD_SK_READV
#undef SESSD_SK_MMA
#undef SESSD_SK_STEP
#undef SESSD_SK_ROUND

    // ---- epilogue: per 32-cout block, M_xi of all 16 xi through LDS, Y = A^T M A per (cout, tile)
    // The arguments only the epilogue needs are re-read from the kernel-argument segment HERE (through an opaque pointer):
    // kept live across the main loop they exhaust the SGPRs, and hipcc then parks a buffer resource in VGPRs and wraps
    // its loads in waterfall loops.
    if (pend_state == 3) {
      SESSD_LDS_BARRIER();
      pend_state = s_pend ? 2 : 0;
    }
    const __attribute__((address_space(4))) WinoArgs* Ep =
        (const __attribute__((address_space(4))) WinoArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(Ep));
    const float* e_scale = Ep->scale ? Ep->scale + (size_t)wset * Ep->ss_stride : nullptr;
    const float* e_shift = Ep->shift ? Ep->shift + (size_t)wset * Ep->ss_stride : nullptr;
    const int e_cout = Ep->cout, e_relu = Ep->relu, e_win = Ep->win, e_tw = Ep->tw, e_ntiles = Ep->ntiles;
    unsigned* e_counter = Ep->counters + u;
    const rsrc_t sr = make_rsrc(Ep->scratch, (unsigned)(2 * G) * SLOT * 4u);
    const bool full = (r0 == 0 && r1 == Ep->rpu);
    float* outb = Ep->out + (size_t)b * e_cout * out_plane;
    const float* resb = Ep->residual ? Ep->residual + (size_t)b * e_cout * out_plane : nullptr;
    // this thread's (cout, tile) pairs: tile tl = tid & 31 in every pass, cout = m_base + cb * 32 + (tid >> 5) + (NT / 32) * n
    const int tl = tid & 31, col0 = tid >> 5;
    int tt = t_base + tl;
    bool tok = tt < e_ntiles;
    size_t pix = 0;
    if constexpr (LIST) {
      const int ee = tt < n_list ? Ep->tile_list[tt] : -1;
      tok = ee >= 0;
      const int eb = tok ? ee / e_ntiles : 0;
      tt = tok ? ee - eb * e_ntiles : 0;
      pix = (size_t)eb * e_cout * out_plane;
    }
    const int oty = tok ? tt / e_tw : 0, otx = tok ? tt - (tt / e_tw) * e_tw : 0;
    pix += (size_t)(2 * oty) * e_win + 2 * otx;
    // parts of unit u = the shares that intersect its rounds; share of round q = ((q + 1) G - 1) / R
    const int w_first = (int)((((long long)u * Ep->rpu + 1) * G - 1) / R);
    const int w_last = (int)((((long long)u * Ep->rpu + Ep->rpu) * G - 1) / R);
    const unsigned my_slot = (unsigned)(2 * w + (r0 == 0 ? 1 : 0)) * (unsigned)(SLOT * 4);
    // A part that finds every other part already counted is the last one for certain: it neither writes nor counts, it
    // fetches the (single) other part now and adds it on the fly.
    bool certain = false;
    unsigned oslot = 0;
    if (!full && w_last == w_first + 1) {
      // one reader: every thread of the workgroup must take the same path
      if (tid == 0) s_last = __hip_atomic_load(e_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u ? 1 : 0;
      __syncthreads();
      certain = s_last != 0;
      const int wq = (w == w_first) ? w_last : w_first;
      oslot = (unsigned)(2 * wq + (wq == w_first ? 1 : 0)) * (unsigned)(SLOT * 4);
    }
    const bool writes_out = full || certain;
    // other-part values and BatchNorm constants of a pass are fetched one pass ahead, into the registers the previous pass's
    // accumulators have just left (fetching all passes up front spills; fetching a pass at its own start exposes one memory
    // latency per pass)
    f32x4v onext[NPT];
    float scn[NPT], shn[NPT];
#define SESSD_SK_FETCH(CB)                                                                          \
  _Pragma("unroll") for (int n = 0; n < NPT; ++n) {                                                 \
    const int co_ = m_base + (CB)*32 + col0 + (NT / 32) * n;                                        \
    if (certain)                                                                                    \
      onext[n] = __builtin_bit_cast(                                                                \
          f32x4v, __builtin_amdgcn_raw_buffer_load_b128(                                            \
                      sr, (int)(oslot + (unsigned)((((CB)*32 + col0 + (NT / 32) * n) * 32 + tl) * 16)), 0, SESSD_SYSTEM_SCOPE)); \
    else                                                                                            \
      onext[n] = 0.f;                                                                               \
    scn[n] = (writes_out && e_scale && co_ < e_cout) ? e_scale[co_] : 1.f;                          \
    shn[n] = (writes_out && e_shift && co_ < e_cout) ? e_shift[co_] : 0.f;                          \
  }
    SESSD_SK_FETCH(0)
#pragma unroll
    for (int cb = 0; cb < CBN; ++cb) {
      if (cb) SESSD_LDS_BARRIER();
#pragma unroll
      for (int x = 0; x < XW; ++x)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int co = (q & 3) + 8 * (q >> 2) + 4 * h;
          lds[((wave * XW + x) * 32 + co) * 32 + j] = acc[x][cb][q];
        }
      f32x4v other[NPT];
      float sc[NPT], sh[NPT];
#pragma unroll
      for (int n = 0; n < NPT; ++n) { other[n] = onext[n]; sc[n] = scn[n]; sh[n] = shn[n]; }
      if (cb + 1 < CBN) { SESSD_SK_FETCH(cb + 1) }
      SESSD_LDS_BARRIER();
#pragma unroll
      for (int n = 0; n < NPT; ++n) {
        const int col = col0 + (NT / 32) * n;
        float m[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) m[x] = lds[(x * 32 + col) * 32 + tl];
        float q0[4], q1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          q0[c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
          q1[c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
        }
        float y[4];
        y[0] = q0[0] + q0[1] + q0[2]; y[1] = q0[1] - q0[2] - q0[3];
        y[2] = q1[0] + q1[1] + q1[2]; y[3] = q1[1] - q1[2] - q1[3];
        const int co = m_base + cb * 32 + col;
        if (full) {
          SESSD_SK_FINALIZE(y, co, sc[n], sh[n])
        } else if (certain) {
          // share order: the part of the lower share first (a + b is commutative; written out for the general rule)
          const f32x4v o4 = other[n];
          const bool mine_first = (w == w_first);
          float z[4];
          z[0] = mine_first ? (0.f + y[0]) + o4.x : (0.f + o4.x) + y[0];
          z[1] = mine_first ? (0.f + y[1]) + o4.y : (0.f + o4.y) + y[1];
          z[2] = mine_first ? (0.f + y[2]) + o4.z : (0.f + o4.z) + y[2];
          z[3] = mine_first ? (0.f + y[3]) + o4.w : (0.f + o4.w) + y[3];
          SESSD_SK_FINALIZE(z, co, sc[n], sh[n])
        } else {
          f32x4v v;
          v.x = y[0]; v.y = y[1]; v.z = y[2]; v.w = y[3];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4g, v), sr,
                                                 (int)(my_slot + (unsigned)(((cb * 32 + col) * 32 + tl) * 16)), 0, SESSD_SYSTEM_SCOPE);
        }
      }
    }
    if (certain) {
      if (tid == 0) __hip_atomic_store(e_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (!full) {
      // My partial outputs are on their way to memory. Counting this part needs them acknowledged: if more work follows, that
      // wait is folded into the next segment's prologue (SESSD_SK_SIGNAL below), and the (unlikely) duty of finishing the unit
      // is carried out after the last segment.
      if (pend_state == 2) SESSD_SK_FINISH_PENDING()
      pend_u = u; pend_first = w_first; pend_last = w_last; pend_tbase = t_base; pend_mbase = m_base; pend_b = b;
      pend_state = 1;  // stores issued, not yet counted
    }
#undef SESSD_SK_FETCH
    SESSD_LDS_BARRIER();  // the next segment's prologue overwrites the LDS the epilogue read
  }
  if (pend_state == 1) SESSD_SK_SIGNAL()
  if (pend_state == 2) SESSD_SK_FINISH_PENDING()
#undef SESSD_SK_FINALIZE
#undef SESSD_SK_SIGNAL
#undef SESSD_SK_FINISH_PENDING
}


// ---------------------------------------------------------------------------------------------------------------------------
// THIRD generation (tile_cfg 24, shape 2): the output transform stays in REGISTERS. From the cost breakdown of the kernel above
// (DESIGN.md section 3): its main loop runs at 89 % of the matrix cores' issue rate, but every segment of a workgroup ends in an
// epilogue that moves the 16 M_xi of each 32-cout block through LDS (64 KB written + 64 KB read per block, four blocks, two
// barriers each) before Y = A^T M A can be formed -- ~5.5 us, twice per workgroup, ~19 % of a 58 us launch that the matrix
// cores sit out. Here a wave owns ALL 16 transform points of its 32 couts:
//   unit      = 32 consecutive 2x2-output tiles x 128 couts x all input channels (as shape 0)
//   workgroup = 4 waves, one per SIMD; wave w owns couts 32 w .. 32 w + 31 for all 16 xi: 16 accumulators of 32x32 (256
//               registers; one wave per SIMD has the whole 512-entry file)
//   round     = 8 k-steps (16 input channels: shape 0's round, so the stream-K cuts fall where shape 0's fall and the bits are
//               the same); wave w transforms the patches of k-steps w and w + 4 of the NEXT round into the other V buffer (two
//               halves of a round, one barrier per round); per k-step 16 MFMAs: B (V_xi) by 16 LDS reads, A (U_xi, packed
//               [cout group][k-step][wave][parity][cout][xi 16]: a lane's 16 operands are four 16-byte loads) through a register
//               ring of 4 sets loaded 3 k-steps ahead
//   epilogue  = per accumulator element (cout row, tile column) the 16 M_xi are 16 registers of ONE lane: Y = A^T M A, BatchNorm,
//               ReLU, residual and the stores without LDS and without a barrier.
// Stream-K bookkeeping (shares, cut units, write-through partial slots, deferred counting) as above. Per (cout, tile) the
// arithmetic is shape 0's in the same order: BIT-IDENTICAL results for the same workgroup count.
template <int VAR = 0>
__global__ __launch_bounds__(256, 1) void conv3x3s1_winograd_rk_kernel(WinoArgs A) {
  constexpr int NW = 4, KR = 8, RING = 4;
  constexpr int NT = NW * 64;
  constexpr unsigned WSTEP = 4u * 2u * 32u * 16u * 4u;  // bytes of packed U per k-step (128 couts x 2 channels x 16 xi)
  constexpr int VBUF = KR * 1024;                        // floats of one V buffer: [ks 8][xi 16][h 2][tile 32]
  constexpr int SLOT = 4 * 32 * 32 * 4;                  // floats of one scratch slot: [cout 128][tile 32][4]
  constexpr int NPT = 1024 / NT;                         // (cout, tile) pairs per thread and 32-cout pass of FINISH_PENDING
  __shared__ __attribute__((aligned(16))) float lds[2 * VBUF];  // 64 KB: the V double buffer
  __shared__ int s_last, s_pend;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int G = gridDim.x;
  const int w = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
  const long long R = A.total_rounds;
  int r = (int)((long long)w * R / G);
  const int r_stop = (int)((long long)(w + 1) * R / G);
  const int in_plane = A.hin * A.win;
  const size_t out_plane = (size_t)in_plane;
  const unsigned xstep = 2u * (unsigned)in_plane * 4u;  // bytes per k-step (2 channels)
  const unsigned wo = (unsigned)(((wave * 2 + h) * 32 + j) * 64);

  int pend_state = 0, pend_u = 0, pend_first = 0, pend_last = 0, pend_tbase = 0, pend_mbase = 0, pend_b = 0;
#define SESSD_RK_SIGNAL()                                                                          \
  {                                                                                                \
    __builtin_amdgcn_s_waitcnt(0);                                                                 \
    __syncthreads();                                                                               \
    if (tid == 0) {                                                                                \
      const unsigned old = __hip_atomic_fetch_add(A.counters + pend_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
      s_pend = (old == (unsigned)(pend_last - pend_first)) ? 1 : 0;                                \
    }                                                                                              \
    __syncthreads();                                                                               \
    pend_state = s_pend ? 2 : 0;                                                                   \
  }
#define SESSD_RK_FINALIZE(Y, CO, SC, SH)                                                            \
  if (tok && (CO) < e_cout) {                                                                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                 \
      const size_t o = (size_t)(CO) * out_plane + pix + (size_t)a * e_win;                          \
      float v0 = fmaf((Y)[2 * a], (SC), (SH)), v1 = fmaf((Y)[2 * a + 1], (SC), (SH));              \
      if (e_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }                                 \
      if (resb) { v0 += resb[o]; v1 += resb[o + 1]; }                                               \
      *reinterpret_cast<float2*>(outb + o) = make_float2(v0, v1);                                   \
    }                                                                                               \
  }
#define SESSD_RK_FINISH_PENDING()                                                                  \
  {                                                                                                \
    const __attribute__((address_space(4))) WinoArgs* Fp =                                         \
        (const __attribute__((address_space(4))) WinoArgs*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(Fp));                                                                   \
    const int f_set = pend_b / Fp->bper;                                                           \
    const float* f_scale = Fp->scale ? Fp->scale + (size_t)f_set * Fp->ss_stride : nullptr;        \
    const float* f_shift = Fp->shift ? Fp->shift + (size_t)f_set * Fp->ss_stride : nullptr;        \
    const int e_cout = Fp->cout, e_relu = Fp->relu, e_win = Fp->win, f_tw = Fp->tw;                \
    const rsrc_t fr = make_rsrc(Fp->scratch, (unsigned)(2 * G) * SLOT * 4u);                       \
    float* outb = Fp->out + (size_t)pend_b * e_cout * out_plane;                                   \
    const float* resb = Fp->residual ? Fp->residual + (size_t)pend_b * e_cout * out_plane : nullptr; \
    const int tl = tid & 31, col0 = tid >> 5;                                                      \
    const int tt = pend_tbase + tl;                                                                \
    const bool tok = tt < Fp->ntiles;                                                              \
    const int oty = tok ? tt / f_tw : 0, otx = tok ? tt - (tt / f_tw) * f_tw : 0;                  \
    const size_t pix = (size_t)(2 * oty) * e_win + 2 * otx;                                        \
    _Pragma("unroll 1") for (int cb = 0; cb < 4; ++cb) {                                           \
      f32x4v ysum[NPT];                                                                            \
      _Pragma("unroll") for (int n = 0; n < NPT; ++n) ysum[n] = 0.f;                               \
      for (int wq = pend_first; wq <= pend_last; ++wq) {                                           \
        const unsigned slot = (unsigned)(2 * wq + (wq == pend_first ? 1 : 0)) * (unsigned)(SLOT * 4); \
        f32x4v p[NPT];                                                                             \
        _Pragma("unroll") for (int n = 0; n < NPT; ++n)                                            \
          p[n] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(                 \
              fr, (int)(slot + (unsigned)(((cb * 32 + col0 + (NT / 32) * n) * 32 + tl) * 16)), 0, SESSD_SYSTEM_SCOPE)); \
        _Pragma("unroll") for (int n = 0; n < NPT; ++n) ysum[n] += p[n];                           \
      }                                                                                            \
      _Pragma("unroll") for (int n = 0; n < NPT; ++n) {                                            \
        const int co = pend_mbase + cb * 32 + col0 + (NT / 32) * n;                                \
        const float scv = (f_scale && co < e_cout) ? f_scale[co] : 1.f, shv = (f_shift && co < e_cout) ? f_shift[co] : 0.f; \
        const float z[4] = {ysum[n].x, ysum[n].y, ysum[n].z, ysum[n].w};                           \
        SESSD_RK_FINALIZE(z, co, scv, shv)                                                         \
      }                                                                                            \
    }                                                                                              \
    if (tid == 0) __hip_atomic_store(Fp->counters + pend_u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    pend_state = 0;                                                                                \
  }

  while (r < r_stop) {
    const int u = __builtin_amdgcn_readfirstlane(r / A.rpu);
    const int r0 = __builtin_amdgcn_readfirstlane(r - u * A.rpu);
    const int r1 = __builtin_amdgcn_readfirstlane(min(r_stop - u * A.rpu, A.rpu));
    r = u * A.rpu + r1;
    const int cg = __builtin_amdgcn_readfirstlane(u % A.ngroups), ub = u / A.ngroups;
    const int tb = __builtin_amdgcn_readfirstlane(ub % A.tblocks), b = __builtin_amdgcn_readfirstlane(ub / A.tblocks);
    const int t_base = tb * 32, m_base = cg * 128;
    const rsrc_t xr = make_rsrc(A.in + (size_t)b * A.cin * in_plane, (unsigned)A.cin * in_plane * 4u);
    const int wset = __builtin_amdgcn_readfirstlane(b / A.bper);
    const rsrc_t wr = make_rsrc(A.upk + (size_t)wset * A.upk_stride + (size_t)cg * (A.cin >> 1) * (WSTEP / 4), (unsigned)(A.cin >> 1) * WSTEP);

    // ---- transform role: lane = (tile j, channel parity h)
    const int t = t_base + j;
    const bool tlive = t < A.ntiles;
    const int ty = tlive ? t / A.tw : 0, tx = tlive ? t - (t / A.tw) * A.tw : 0;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    unsigned ro[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = y0 + q;
      ro[q] = (tlive && y >= 0 && y < A.hin) ? (unsigned)((h * in_plane + y * A.win + max(x0, 0)) * 4) : SESSD_OOB;
    }
    const bool mask_l = (tx == 0), mask_r = (tx == A.tw - 1);
    const bool edge = __builtin_amdgcn_ballot_w64(tlive && (mask_l || mask_r)) != 0;
    const int klast = r1 * KR - 1;

    f32x16 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[x][q] = 0.f;
    f32x4v pr[4];
    f32x4v ua[RING][4];
    float bv[2][16];

#define SESSD_RK_LOADP(ROUND, HALF)                                                                \
  {                                                                                                \
    const unsigned xs = (unsigned)(min((ROUND), r1 - 1) * KR + wave + 4 * (HALF)) * xstep;         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
      pr[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)ro[q], (int)xs, 0)); \
  }
#define SESSD_RK_LOADU(SET, KG)                                                                    \
  {                                                                                                \
    const unsigned ws = (unsigned)min((KG), klast) * WSTEP;                                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                  \
      ua[SET][e] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(wo + 16u * e), (int)ws, 0)); \
  }
#define SESSD_RK_TRANSFORM(VOFF, HALF)                                                             \
  {                                                                                                \
    if (edge) {                                                                                    \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
        const f32x4v p = pr[q];                                                                    \
        pr[q].x = mask_l ? 0.f : p.x; pr[q].y = mask_l ? p.x : p.y;                                \
        pr[q].z = mask_l ? p.y : p.z; pr[q].w = mask_l ? p.z : (mask_r ? 0.f : p.w);               \
      }                                                                                            \
    }                                                                                              \
    f32x2v tl[4], tr[4];                                                                           \
    tl[0] = pr[0].xy - pr[2].xy; tr[0] = pr[0].zw - pr[2].zw;                                      \
    tl[1] = pr[1].xy + pr[2].xy; tr[1] = pr[1].zw + pr[2].zw;                                      \
    tl[2] = pr[2].xy - pr[1].xy; tr[2] = pr[2].zw - pr[1].zw;                                      \
    tl[3] = pr[1].xy - pr[3].xy; tr[3] = pr[1].zw - pr[3].zw;                                      \
    float* dst = &lds[(VOFF) + (wave + 4 * (HALF)) * 1024 + h * 32 + j];                           \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                \
      dst[(a * 4 + 0) * 64] = tl[a].x - tr[a].x;                                                   \
      dst[(a * 4 + 1) * 64] = tl[a].y + tr[a].x;                                                   \
      dst[(a * 4 + 2) * 64] = tr[a].x - tl[a].y;                                                   \
      dst[(a * 4 + 3) * 64] = tl[a].y - tr[a].y;                                                   \
    }                                                                                              \
  }
#define SESSD_RK_READV(P, KS, VOFF)                                                                \
  {                                                                                                \
    const float* vb = &lds[(VOFF) + (KS)*1024 + h * 32 + j];                                       \
    _Pragma("unroll") for (int x = 0; x < 16; ++x) bv[P][x] = vb[x * 64];                          \
  }
#define SESSD_RK_MMA(SET, P)                                                                       \
  {                                                                                                \
    _Pragma("unroll") for (int x = 0; x < 16; ++x)                                                 \
      acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[SET][x >> 2][x & 3], bv[P][x], acc[x], 0, 0, 0); \
  }
#define SESSD_RK_STEP(KS)                                                                          \
  {                                                                                                \
    SESSD_RK_LOADU(((KS) + RING - 1) & (RING - 1), kg0 + (KS) + RING - 1)                          \
    if ((KS) < KR - 1) SESSD_RK_READV(((KS) + 1) & 1, (KS) + 1, voff)                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_RK_MMA((KS) & (RING - 1), (KS) & 1)                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
#define SESSD_RK_ROUND()                                                                           \
  {                                                                                                \
    const int kg0 = rr * KR;                                                                       \
    SESSD_RK_LOADP(rr + 1, 0)                                                                      \
    SESSD_RK_READV(0, 0, voff)                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_RK_STEP(0)                                                                               \
    SESSD_RK_STEP(1)                                                                               \
    SESSD_RK_STEP(2)                                                                               \
    SESSD_RK_STEP(3)                                                                               \
    SESSD_RK_TRANSFORM(voff ^ VBUF, 0)                                                             \
    SESSD_RK_LOADP(rr + 1, 1)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_RK_STEP(4)                                                                               \
    SESSD_RK_STEP(5)                                                                               \
    SESSD_RK_STEP(6)                                                                               \
    SESSD_RK_STEP(7)                                                                               \
    SESSD_RK_TRANSFORM(voff ^ VBUF, 1)                                                             \
    __syncthreads();                                                                               \
    voff ^= VBUF;                                                                                  \
    ++rr;                                                                                          \
  }

    SESSD_RK_LOADP(r0, 0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) {  // in ring order
      SESSD_RK_LOADU(s, r0 * KR + s)
      __builtin_amdgcn_sched_barrier(0);
    }
    SESSD_RK_TRANSFORM(0, 0)
    SESSD_RK_LOADP(r0, 1)
    __builtin_amdgcn_sched_barrier(0);
    SESSD_RK_TRANSFORM(0, 1)
    __syncthreads();
    if (pend_state == 1) {
      // the previous segment's part: its stores are older than the patch loads every thread has just consumed
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(A.counters + pend_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_pend = (old == (unsigned)(pend_last - pend_first)) ? 1 : 0;
      }
      pend_state = 3;  // counted; the answer is read at this segment's epilogue
    }
    int voff = 0;
    int rr = r0;
    while (rr < r1) SESSD_RK_ROUND()
#undef SESSD_RK_LOADP
#undef SESSD_RK_LOADU
#undef SESSD_RK_TRANSFORM
#undef SESSD_RK_READV
#undef SESSD_RK_MMA
#undef SESSD_RK_STEP
#undef SESSD_RK_ROUND

    // ---- epilogue: everything of a (cout, tile) pair is in ONE lane
    if (pend_state == 3) {
      SESSD_LDS_BARRIER();
      pend_state = s_pend ? 2 : 0;
    }
    const __attribute__((address_space(4))) WinoArgs* Ep =
        (const __attribute__((address_space(4))) WinoArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(Ep));
    const float* e_scale = Ep->scale ? Ep->scale + (size_t)wset * Ep->ss_stride : nullptr;
    const float* e_shift = Ep->shift ? Ep->shift + (size_t)wset * Ep->ss_stride : nullptr;
    const int e_cout = Ep->cout, e_relu = Ep->relu, e_win = Ep->win, e_tw = Ep->tw, e_ntiles = Ep->ntiles;
    unsigned* e_counter = Ep->counters + u;
    const rsrc_t sr = make_rsrc(Ep->scratch, (unsigned)(2 * G) * SLOT * 4u);
    const bool full = (r0 == 0 && r1 == Ep->rpu);
    float* outb = Ep->out + (size_t)b * e_cout * out_plane;
    const float* resb = Ep->residual ? Ep->residual + (size_t)b * e_cout * out_plane : nullptr;
    // this lane's pairs: tile j, couts m_base + wave * 32 + (q & 3) + 8 (q >> 2) + 4 h for the 16 accumulator elements q
    const int tt = t_base + j;
    const bool tok = tt < e_ntiles;
    const int oty = tok ? tt / e_tw : 0, otx = tok ? tt - (tt / e_tw) * e_tw : 0;
    const size_t pix = (size_t)(2 * oty) * e_win + 2 * otx;
    const int w_first = (int)((((long long)u * Ep->rpu + 1) * G - 1) / R);
    const int w_last = (int)((((long long)u * Ep->rpu + Ep->rpu) * G - 1) / R);
    const unsigned my_slot = (unsigned)(2 * w + (r0 == 0 ? 1 : 0)) * (unsigned)(SLOT * 4);
    bool certain = false;
    unsigned oslot = 0;
    if (!full && w_last == w_first + 1) {
      if (tid == 0) s_last = __hip_atomic_load(e_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u ? 1 : 0;
      __syncthreads();
      certain = s_last != 0;
      const int wq = (w == w_first) ? w_last : w_first;
      oslot = (unsigned)(2 * wq + (wq == w_first ? 1 : 0)) * (unsigned)(SLOT * 4);
    }
    const bool writes_out = full || certain;
    const int col_base = wave * 32 + 4 * h;   // + (q & 3) + 8 (q >> 2)
    // eight pairs at a time: their BatchNorm constants and the other part's values are requested together (one memory latency)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float sc[8], sh[8];
      f32x4v other[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = half * 8 + e;
        const int col = col_base + (q & 3) + 8 * (q >> 2);
        const int co_ = m_base + col;
        sc[e] = (writes_out && e_scale && co_ < e_cout) ? e_scale[co_] : 1.f;
        sh[e] = (writes_out && e_shift && co_ < e_cout) ? e_shift[co_] : 0.f;
        if (certain)
          other[e] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                                                    sr, (int)(oslot + (unsigned)((col * 32 + j) * 16)), 0, SESSD_SYSTEM_SCOPE));
        else
          other[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = half * 8 + e;
        const int col = col_base + (q & 3) + 8 * (q >> 2);
        float q0[4], q1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          q0[c] = acc[0 * 4 + c][q] + acc[1 * 4 + c][q] + acc[2 * 4 + c][q];
          q1[c] = acc[1 * 4 + c][q] - acc[2 * 4 + c][q] - acc[3 * 4 + c][q];
        }
        float y[4];
        y[0] = q0[0] + q0[1] + q0[2]; y[1] = q0[1] - q0[2] - q0[3];
        y[2] = q1[0] + q1[1] + q1[2]; y[3] = q1[1] - q1[2] - q1[3];
        const int co = m_base + col;
        if (full) {
          SESSD_RK_FINALIZE(y, co, sc[e], sh[e])
        } else if (certain) {
          const f32x4v o4 = other[e];
          const bool mine_first = (w == w_first);
          float z[4];
          z[0] = mine_first ? (0.f + y[0]) + o4.x : (0.f + o4.x) + y[0];
          z[1] = mine_first ? (0.f + y[1]) + o4.y : (0.f + o4.y) + y[1];
          z[2] = mine_first ? (0.f + y[2]) + o4.z : (0.f + o4.z) + y[2];
          z[3] = mine_first ? (0.f + y[3]) + o4.w : (0.f + o4.w) + y[3];
          SESSD_RK_FINALIZE(z, co, sc[e], sh[e])
        } else {
          f32x4v v;
          v.x = y[0]; v.y = y[1]; v.z = y[2]; v.w = y[3];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4g, v), sr, (int)(my_slot + (unsigned)((col * 32 + j) * 16)), 0,
                                                 SESSD_SYSTEM_SCOPE);
        }
      }
    }
    if (certain) {
      if (tid == 0) __hip_atomic_store(e_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (!full) {
      if (pend_state == 2) SESSD_RK_FINISH_PENDING()
      pend_u = u; pend_first = w_first; pend_last = w_last; pend_tbase = t_base; pend_mbase = m_base; pend_b = b;
      pend_state = 1;  // stores issued, not yet counted
    }
    // (no LDS was touched by the epilogue: the next segment's prologue may overwrite the V buffers at once; the main loop's last
    // barrier already ordered every wave's last V reads before it)
  }
  if (pend_state == 1) SESSD_RK_SIGNAL()
  if (pend_state == 2) SESSD_RK_FINISH_PENDING()
#undef SESSD_RK_FINALIZE
#undef SESSD_RK_SIGNAL
#undef SESSD_RK_FINISH_PENDING
}

int launch_rk(const float* in, int batch, int nsets, int cin, int h, int w, const float* upk, float* out, int cout, const float* scale,
              const float* shift, int relu, const float* residual, void* workspace, size_t workspace_bytes, int workgroups,
              hipStream_t stream) {
  constexpr int SLOT = 4 * 32 * 32 * 4;
  if (cin % 16) return SESSD_EINVAL;
  WinoArgs A;
  A.in = in; A.upk = upk; A.out = out; A.scale = scale; A.shift = shift; A.residual = residual;
  A.cin = cin; A.hin = h; A.win = w; A.cout = cout; A.relu = relu;
  A.tw = w / 2; A.ntiles = (h / 2) * (w / 2); A.tblocks = sessd_divup(A.ntiles, 32); A.ngroups = sessd_divup(cout, 128);
  A.rpu = cin / 16;
  A.bper = batch / nsets; A.ss_stride = nsets > 1 ? cout : 0;
  A.upk_stride = nsets > 1 ? (long long)sessd_divup(cout, 128) * (cin >> 1) * (4 * 2 * 32 * 16) : 0;
  A.tile_list = nullptr; A.n_list = nullptr; A.list_cap = 0; A.min_rounds = 1; A.batch = batch;
  const long long units = (long long)batch * A.tblocks * A.ngroups;
  if (units * A.rpu > 0x7fffffffLL) return SESSD_EINVAL;
  A.total_rounds = (int)(units * A.rpu);
  const size_t need = sessd_align((size_t)units * 4, 256) + (size_t)2 * workgroups * SLOT * 4;
  if (need > workspace_bytes) return SESSD_EWORKSPACE;
  if (workgroups > A.total_rounds) workgroups = A.total_rounds >= 8 ? (A.total_rounds & ~7) : A.total_rounds;
  A.counters = (unsigned*)workspace;
  A.scratch = (float*)((char*)workspace + sessd_align((size_t)units * 4, 256));
  SESSD_LAUNCH((conv3x3s1_winograd_rk_kernel<0>), dim3(workgroups), dim3(256), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

template <int NW, int CBN>
int launch_sk(const float* in, int batch, int nsets, int cin, int h, int w, const float* upk, float* out, int cout, const float* scale,
              const float* shift, int relu, const float* residual, void* workspace, size_t workspace_bytes, int workgroups,
              hipStream_t stream, const int* tile_list = nullptr, const int* n_list = nullptr, int list_cap = 0, int min_rounds = 1) {
  constexpr int SLOT = CBN * 32 * 32 * 4;
  if (cin % (2 * NW)) return SESSD_EINVAL;
  if (tile_list && (nsets != 1 || !n_list || list_cap < 1 || (size_t)batch * cin * h * w * 4 >= 0xFFFFFFFFull)) return SESSD_EINVAL;
  WinoArgs A;
  A.in = in; A.upk = upk; A.out = out; A.scale = scale; A.shift = shift; A.residual = residual;
  A.cin = cin; A.hin = h; A.win = w; A.cout = cout; A.relu = relu;
  A.tw = w / 2; A.ntiles = (h / 2) * (w / 2); A.tblocks = sessd_divup(A.ntiles, 32); A.ngroups = sessd_divup(cout, CBN * 32);
  A.rpu = cin / (2 * NW);
  A.bper = batch / nsets; A.ss_stride = nsets > 1 ? cout : 0;
  A.upk_stride = nsets > 1 ? (long long)sessd_divup(cout, CBN * 32) * (cin >> 1) * (NW * 2 * 32 * 32 / 4) : 0;
  // min_rounds < 0 (list launches only): whole-unit shares, at least -min_rounds units per workgroup
  A.tile_list = tile_list; A.n_list = n_list; A.list_cap = list_cap; A.min_rounds = (min_rounds < 0 && tile_list) ? min_rounds : (min_rounds < 1 ? 1 : min_rounds); A.batch = batch;
  const long long units = (long long)batch * A.tblocks * A.ngroups;
  if (units * A.rpu > 0x7fffffffLL) return SESSD_EINVAL;
  A.total_rounds = (int)(units * A.rpu);
  const size_t need = sessd_align((size_t)units * 4, 256) + (size_t)2 * workgroups * SLOT * 4;
  if (need > workspace_bytes) return SESSD_EWORKSPACE;
  // every share must hold at least one round (the part count of a cut unit is a difference of share indices)
  if (workgroups > A.total_rounds) workgroups = A.total_rounds >= 8 ? (A.total_rounds & ~7) : A.total_rounds;
  A.counters = (unsigned*)workspace;
  A.scratch = (float*)((char*)workspace + sessd_align((size_t)units * 4, 256));
  if (tile_list) {   // active-tile mode: the same kernel over a device list of tiles (shares are sized on the device)
    SESSD_LAUNCH((conv3x3s1_winograd_sk_kernel<NW, CBN, 0, true>), dim3(workgroups), dim3(NW * 64), 0, stream, A);
    SESSD_CHECK_LAUNCH();
    return SESSD_OK;
  }
  static const int var = [] { const char* e = getenv("SESSD_WINO_VAR"); return e ? atoi(e) & 3 : 0; }();
  switch (var) {
    case 1: SESSD_LAUNCH((conv3x3s1_winograd_sk_kernel<NW, CBN, 1>), dim3(workgroups), dim3(NW * 64), 0, stream, A); break;
    case 2: SESSD_LAUNCH((conv3x3s1_winograd_sk_kernel<NW, CBN, 2>), dim3(workgroups), dim3(NW * 64), 0, stream, A); break;
    case 3: SESSD_LAUNCH((conv3x3s1_winograd_sk_kernel<NW, CBN, 3>), dim3(workgroups), dim3(NW * 64), 0, stream, A); break;
    default: SESSD_LAUNCH((conv3x3s1_winograd_sk_kernel<NW, CBN, 0>), dim3(workgroups), dim3(NW * 64), 0, stream, A); break;
  }
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int default_workgroups(int shape, int* out) {
  int dev = 0, cus = 0;
  SESSD_TRY(hipGetDevice(&dev));
  SESSD_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  *out = ((shape == 1 ? 2 : 1) * cus) & ~7;
  return *out >= 8 ? SESSD_OK : SESSD_EINVAL;
}

}  // namespace

extern "C" {

// Scratch + counters of sessd_conv3x3_winograd_sk. shape 0: 8 waves x 128 couts per workgroup (one per CU), shape 1: 4 waves x
// 64 couts (two per CU); workgroups 0 = that default. The caller zeroes the workspace ONCE (the kernel leaves the counters
// zero) and must not share it between launches that may run concurrently.
size_t sessd_conv3x3_winograd_sk_workspace_bytes(int batch, int h, int w, int cout, int shape, int workgroups) {
  if (batch < 1 || h < 2 || w < 2 || cout < 1 || workgroups < 0 || shape < 0 || shape > 2) return 0;
  if (workgroups == 0 && default_workgroups(shape, &workgroups) != SESSD_OK) return 0;
  const int cpu = shape == 1 ? 64 : 128;
  const size_t units = (size_t)batch * sessd_divup((h / 2) * (w / 2), 32) * sessd_divup(cout, cpu);
  return sessd_align(units * 4, 256) + (size_t)2 * workgroups * (cpu * 32 * 4) * 4;
}

// Conv2d(cin, cout, 3, stride 1, padding 1) + folded BatchNorm + ReLU + residual, fused Winograd F(2x2,3x3), stream-K over
// `workgroups` persistent workgroups (a multiple of 8; 0 = the shape's default).
// upk = U = G g G^T packed [ceil(cout / C)][cin/2][NW][2][32][C/32][16/NW] with (NW, C) = (8, 128) for shape 0, (4, 64) for
// shape 1 (ops.pack_winograd_sk); even H, W; cin % (2 NW) == 0.
// Several layers of ONE shape in one launch (conv_0 / conv_1 of the SSFA neck, rpn_v1.py:201-210): the batch dimension is
// nsets consecutive groups of batch / nsets elements, group s convolved with weight set s -- upk = nsets packings back to back,
// scale / shift = nsets x cout. One round list, one pipeline fill and one tail instead of nsets.
int sessd_conv3x3_winograd_sk_sets(const float* in, int batch, int nsets, int cin, int h, int w, const float* upk, float* out,
                                   int cout, const float* scale, const float* shift, int relu, const float* residual,
                                   void* workspace, size_t workspace_bytes, int shape, int workgroups, hipStream_t stream) {
  if ((h & 1) || (w & 1) || batch < 1 || nsets < 1 || batch % nsets || cout < 1 || workgroups < 0 || (workgroups & 7) || shape < 0 ||
      shape > 2)
    return SESSD_EINVAL;
  if (workgroups == 0) {
    const int rc = default_workgroups(shape, &workgroups);
    if (rc != SESSD_OK) return rc;
  }
  if (shape == 2)   // third generation: output transform in registers (same cuts and bits as shape 0)
    return launch_rk(in, batch, nsets, cin, h, w, upk, out, cout, scale, shift, relu, residual, workspace, workspace_bytes, workgroups, stream);
  if (shape == 1)
    return launch_sk<4, 2>(in, batch, nsets, cin, h, w, upk, out, cout, scale, shift, relu, residual, workspace, workspace_bytes, workgroups, stream);
  return launch_sk<8, 4>(in, batch, nsets, cin, h, w, upk, out, cout, scale, shift, relu, residual, workspace, workspace_bytes, workgroups, stream);
}

// ACTIVE-TILE mode of the same kernel (round 4; the first layers of the SSFA neck, rpn_v1.py:135-160, on a BEV map that is zero
// outside the sparse backbone's sites): only the 2x2-output tiles listed in tile_list[0 .. min(*n_list, list_cap)) are computed
// and written -- entries image * (h/2 * w/2) + tile in any fixed order, count on the device (sessd_bev_tile_activity builds both).
// The other tiles of `out` are the caller's (sessd_fill_inactive_tiles writes the layer's constant there). Workgroups beyond
// rounds / min_rounds leave at once. min_rounds = -k (round 5): WHOLE-UNIT shares -- every workgroup takes at least k whole
// units (32 tiles x the shape's couts x all input channels), nothing is cut, the scratch slots stay untouched. Same packed U, same workspace (sized for the dense layer) as sessd_conv3x3_winograd_sk.
int sessd_conv3x3_winograd_sk_active(const float* in, int batch, int cin, int h, int w, const float* upk, float* out, int cout,
                                     const float* scale, const float* shift, int relu, const float* residual,
                                     const int32_t* tile_list, const int32_t* n_list, int list_cap, int min_rounds, void* workspace,
                                     size_t workspace_bytes, int shape, int workgroups, hipStream_t stream) {
  if ((h & 1) || (w & 1) || batch < 1 || cout < 1 || workgroups < 0 || (workgroups & 7) || shape < 0 || shape > 1 || !tile_list ||
      !n_list || list_cap < 1)
    return SESSD_EINVAL;
  if (workgroups == 0) {
    const int rc = default_workgroups(shape, &workgroups);
    if (rc != SESSD_OK) return rc;
  }
  if (shape == 1)
    return launch_sk<4, 2>(in, batch, 1, cin, h, w, upk, out, cout, scale, shift, relu, residual, workspace, workspace_bytes, workgroups,
                           stream, tile_list, n_list, list_cap, min_rounds);
  return launch_sk<8, 4>(in, batch, 1, cin, h, w, upk, out, cout, scale, shift, relu, residual, workspace, workspace_bytes, workgroups,
                         stream, tile_list, n_list, list_cap, min_rounds);
}

int sessd_conv3x3_winograd_sk(const float* in, int batch, int cin, int h, int w, const float* upk, float* out, int cout,
                              const float* scale, const float* shift, int relu, const float* residual, void* workspace,
                              size_t workspace_bytes, int shape, int workgroups, hipStream_t stream) {
  return sessd_conv3x3_winograd_sk_sets(in, batch, 1, cin, h, w, upk, out, cout, scale, shift, relu, residual, workspace,
                                        workspace_bytes, shape, workgroups, stream);
}

}  // extern "C"
