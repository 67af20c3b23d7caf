// Sparse 3-D convolution with a FOUR-deep operand ring (variant of sparse_conv.hip).
//
// Same tiling, same fragment layouts, same arithmetic and therefore bit-identical results as sparse_conv_kernel: one wave owns
// 16 output sites x NTW 16-wide cout tiles and walks the tile's active kernel offsets. The difference is how far the loads run
// ahead of the MFMAs. At batch 1 a level has about two waves per SIMD and a wave walks ~20 offsets; with the operands of one
// offset in flight (sparse_conv.hip: index two ahead, rows and weights one ahead) every step costs a full L2/HBM round trip
// (25 us / 21 steps = 1.2 us). Here the gathered rows and weights of THREE offsets are in flight while the fourth is
// multiplied, and the rulebook indices run three further ahead:
//     indices of offsets j+4 .. j+6 | rows + weights of offsets j+1 .. j+3 (register sets) | MFMAs of offset j
// All loads are unconditional (an exhausted offset list re-loads its last offset, a missing neighbour gets an out-of-range
// buffer offset and reads zeros) so that the compiler's vmcnt bookkeeping stays exact.
// Register budget per set: CIN/4 * (1 + NTW) VGPRs -- instantiated only where four sets fit (NTW * CIN <= 128).
//
// STATUS: written in round 1 after the GPU budget was spent -- compiled, not yet run on hardware. Nothing calls it by default;
// tests/test_sparse_conv_deep_gpu.py (SESSD_EXPERIMENTAL=1) checks bit-equality with sessd_sparse_conv.
#include "common.hpp"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bufload1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 bufload4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
#define SESSD_OOB 0x80000000u

template <int CIN, int COUT, int NTW, bool DENSE_OUT>
__global__ __launch_bounds__(256) void sparse_conv_deep_kernel(const float* __restrict__ in_feat,
                                                                const int* __restrict__ nbr,
                                                                const uint32_t* __restrict__ tile_mask, int kv,
                                                                const int* __restrict__ n_dev, int n_cap,
                                                                const float* __restrict__ wpk,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int relu,
                                                                float* __restrict__ out_feat,
                                                                const int* __restrict__ out_indices,
                                                                float* __restrict__ dense_out, int dD, int dH, int dW) {
  constexpr int STEPS = CIN / 4;
  constexpr int NTILE = NTW;
  constexpr int NTALL = COUT / 16;
  constexpr int G = STEPS < 4 ? STEPS : 4;
  constexpr int SG = STEPS / G;
  const int tbase = blockIdx.y * NTW;
  const int lane = threadIdx.x & 63;
  const int tile = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int n = min(n_dev[0], n_cap);
  if (tile * 16 >= n) return;
  const int i = lane & 15, kq = lane >> 4;
  const uint32_t tmask = tile_mask[tile];

  f32x4 acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const rsrc_t fr = make_rsrc(in_feat, 0x7FFFFFFFu);
  const rsrc_t wrs = make_rsrc(wpk, (unsigned)kv * NTALL * STEPS * 64u * 4u);
  float a[4][STEPS], bw[4][NTILE][STEPS];
  uint32_t rest = tmask;
  int remaining = __builtin_popcount(tmask);
  // lanes of the last tile whose site is >= n read the tile's first site (results discarded below): see sparse_conv.hip
  const int* nb = nbr + tile * 16 + (tile * 16 + i < n ? i : 0);
  int klast = 0;
#define SESSD_NEXTK() (rest ? (klast = __builtin_ctz(rest), rest &= rest - 1, klast) : klast)
#define SESSD_LOADAB(SET, K, ROW)                                                                  \
  {                                                                                                \
    const unsigned ao = (ROW) >= 0 ? (unsigned)(((ROW)*CIN + kq * STEPS) * 4) : SESSD_OOB;          \
    const unsigned ws = (unsigned)(K) * (NTALL * STEPS * 64 * 4);                                  \
    if (G == 4) {                                                                                  \
      _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                             \
        const f32x4 v = bufload4(fr, ao + 16u * g, 0);                                             \
        a[SET][4 * g] = v.x; a[SET][4 * g + 1] = v.y; a[SET][4 * g + 2] = v.z; a[SET][4 * g + 3] = v.w; \
      }                                                                                            \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                           \
          const f32x4 v = bufload4(wrs, (unsigned)lane * 16u + (unsigned)((tbase + t) * SG + g) * 1024u, ws); \
          bw[SET][t][4 * g] = v.x; bw[SET][t][4 * g + 1] = v.y; bw[SET][t][4 * g + 2] = v.z; bw[SET][t][4 * g + 3] = v.w; \
        }                                                                                          \
    } else {                                                                                       \
      _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2) a[SET][s2] = bufload1(fr, ao + 4u * s2, 0); \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2)                                       \
          bw[SET][t][s2] = bufload1(wrs, ((unsigned)((tbase + t) * SG) * 64u + lane) * (G * 4u) + 4u * s2, ws); \
    }                                                                                              \
  }
#define SESSD_MMA(SET)                                                                             \
  {                                                                                                \
    _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2)                                           \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[SET][s2], bw[SET][t][s2], acc[t], 0, 0, 0); \
  }
// one step: refill the set that was multiplied last with the head of the index queue, advance the queue, multiply set CUR
#define SESSD_STEP(CUR, FREE)                                                                      \
  {                                                                                                \
    SESSD_LOADAB(FREE, kA, rA)                                                                     \
    kA = kB; rA = rB; kB = kC; rB = rC;                                                            \
    kC = SESSD_NEXTK();                                                                            \
    rC = nb[(size_t)kC * n_cap];                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_MMA(CUR)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
  if (remaining > 0) {
    const int k0 = SESSD_NEXTK();
    const int k1 = SESSD_NEXTK();
    const int k2 = SESSD_NEXTK();
    int kA = SESSD_NEXTK();
    int kB = SESSD_NEXTK();
    int kC = SESSD_NEXTK();
    const int r0 = nb[(size_t)k0 * n_cap];
    const int r1 = nb[(size_t)k1 * n_cap];
    const int r2 = nb[(size_t)k2 * n_cap];
    int rA = nb[(size_t)kA * n_cap];
    int rB = nb[(size_t)kB * n_cap];
    int rC = nb[(size_t)kC * n_cap];
    SESSD_LOADAB(0, k0, r0)
    SESSD_LOADAB(1, k1, r1)
    SESSD_LOADAB(2, k2, r2)
    while (true) {
      SESSD_STEP(0, 3)
      if (--remaining == 0) break;
      SESSD_STEP(1, 0)
      if (--remaining == 0) break;
      SESSD_STEP(2, 1)
      if (--remaining == 0) break;
      SESSD_STEP(3, 2)
      if (--remaining == 0) break;
    }
  }
#undef SESSD_STEP
#undef SESSD_NEXTK
#undef SESSD_LOADAB
#undef SESSD_MMA

  // C/D layout: column (cout) = lane & 15, rows (sites) = (lane >> 4) * 4 + r
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int co = (tbase + t) * 16 + i;
    const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int site = tile * 16 + kq * 4 + r;
      if (site >= n) continue;
      float v = fmaf(acc[t][r], sc, sh);
      if (relu) v = fmaxf(v, 0.f);
      if (DENSE_OUT) {
        const int4 c = *reinterpret_cast<const int4*>(out_indices + (size_t)site * 4);
        dense_out[(((size_t)c.x * COUT + co) * dD + c.y) * dH * dW + (size_t)c.z * dW + c.w] = v;
      } else {
        out_feat[(size_t)site * COUT + co] = v;
      }
    }
  }
}

template <int CIN, int COUT, int NTW>
int launch_ntw(bool dense, const float* in_feat, const int* nbr, const uint32_t* tile_mask, int kv, const int* n_dev,
               int n_cap, const float* wpk, const float* scale, const float* shift, int relu, float* out_feat,
               const int* out_indices, float* dense_out, const int* dd, hipStream_t stream) {
  static_assert(NTW * CIN <= 128, "four operand sets must fit the register file");
  const int tiles = sessd_divup(n_cap, 16);
  dim3 grid(sessd_divup(tiles, 4), COUT / 16 / NTW), block(256);
  if (dense)
    SESSD_LAUNCH((sparse_conv_deep_kernel<CIN, COUT, NTW, true>), grid, block, 0, stream, in_feat, nbr, tile_mask, kv, n_dev,
                 n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, dd[0], dd[1], dd[2]);
  else
    SESSD_LAUNCH((sparse_conv_deep_kernel<CIN, COUT, NTW, false>), grid, block, 0, stream, in_feat, nbr, tile_mask, kv, n_dev,
                 n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, 0, 0, 0);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// the widest cout tile count per wave that keeps four operand sets in registers, not wider than the requested split allows
template <int CIN, int COUT>
int launch(int split, bool dense, const float* in_feat, const int* nbr, const uint32_t* tile_mask, int kv, const int* n_dev,
           int n_cap, const float* wpk, const float* scale, const float* shift, int relu, float* out_feat,
           const int* out_indices, float* dense_out, const int* dd, hipStream_t stream) {
  constexpr int NT = COUT / 16;
  if (split <= 0) split = (n_cap / 16 < 4096) ? (NT >= 4 ? 4 : (NT >= 2 ? 2 : 1)) : 1;
#define SESSD_ARGS dense, in_feat, nbr, tile_mask, kv, n_dev, n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, dd, stream
  if constexpr (NT % 4 == 0) {
    if (split >= 4 || (NT / 2) * CIN > 128) return launch_ntw<CIN, COUT, NT / 4>(SESSD_ARGS);
  }
  if constexpr (NT % 2 == 0 && (NT / 2) * CIN <= 128) {
    if (split >= 2 || NT * CIN > 128) return launch_ntw<CIN, COUT, NT / 2>(SESSD_ARGS);
  }
  if constexpr (NT * CIN <= 128) return launch_ntw<CIN, COUT, NT>(SESSD_ARGS);
#undef SESSD_ARGS
  return SESSD_EINVAL;
}

}  // namespace

extern "C" {

// Same contract as sessd_sparse_conv (same packed weights, rulebook and outputs; results bit-identical); channel pairs of
// SpMiddleFHD only. EXPERIMENTAL -- not yet validated on hardware.
int sessd_sparse_conv_deep(const float* in_feat, int cin, const int* nbr, const uint32_t* tile_mask, int kernel_volume,
                           const int* n_out_dev, int n_out_cap, const float* packed_weight, const float* scale,
                           const float* shift, int relu, float* out_feat, int cout, const int* out_indices, float* dense_out,
                           const int* dense_dims3, int cout_split, hipStream_t stream) {
  if (n_out_cap <= 0 || kernel_volume <= 0 || kernel_volume > 32) return SESSD_EINVAL;
  const bool dense = dense_out != nullptr;
  if (dense && (!out_indices || !dense_dims3)) return SESSD_EINVAL;
  if (!dense && !out_feat) return SESSD_EINVAL;
#define SESSD_SC(CI, CO)                                                                                                     \
  if (cin == CI && cout == CO)                                                                                               \
    return launch<CI, CO>(cout_split, dense, in_feat, nbr, tile_mask, kernel_volume, n_out_dev, n_out_cap, packed_weight, scale, \
                          shift, relu, out_feat, out_indices, dense_out, dense_dims3, stream);
  SESSD_SC(4, 16)
  SESSD_SC(16, 16)
  SESSD_SC(16, 32)
  SESSD_SC(32, 32)
  SESSD_SC(32, 64)
  SESSD_SC(64, 64)
#undef SESSD_SC
  return SESSD_EINVAL;
}

}  // extern "C"
