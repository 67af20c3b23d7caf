// Backward of the sparse 3-D convolution on gfx950 ("sparse-conv backward: dgrad via transposed
// rulebook, wgrad"). The forward is spconv's indice_conv as called from det3d/models/backbones/scn.py:106-148; the
// SE-SSD training step differentiates it through trainer_sessd.py:250-275 (student forward + backward).
//
//   y[j] = sum_k W_k^T x[nbr[k][j]]                     (forward, sparse_conv.hip)
//   dx[i] = sum_k W_k  dy[nbrT[k][i]]                   dgrad: the SAME output-stationary kernel, run over the input
//                                                       sites with the transposed rulebook nbrT[k][i] = j <=> nbr[k][j] = i
//                                                       and per-offset transposed weights (host packs W_k^T)
//   dW_k[ci][co] = sum_j x[nbr[k][j]][ci] dy[j][co]     wgrad: one GEMM per offset with the SITES as the reduction axis
//
// This file holds the rulebook transpose and the wgrad kernels. Both are deterministic: a (k, i) pair has exactly one
// j (an input site and an offset determine the output cell), so the transpose is a collision-free scatter; wgrad sums
// each site chunk in order on the matrix cores and then the <= 64 chunk partials in order.
#include "common.hpp"

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bufload1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
#define SESSD_OOB 0x80000000u
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_CHUNKS = 64;  // site chunks per offset (partials reduced in order afterwards)

// grid (ceil(n_out_cap / 256), kv): nbr_t / tile_mask_t pre-filled with -1 / 0
__global__ __launch_bounds__(256) void rulebook_transpose_kernel(const int* __restrict__ nbr,
                                                                  const int* __restrict__ n_out_dev, int n_out_cap,
                                                                  int n_in_cap, int* __restrict__ nbr_t,
                                                                  uint32_t* __restrict__ tile_mask_t) {
  const int j = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (j >= min(n_out_dev[0], n_out_cap)) return;
  const int i = nbr[(size_t)k * n_out_cap + j];
  if (i < 0 || i >= n_in_cap) return;
  nbr_t[(size_t)k * n_in_cap + i] = j;
  atomicOr(&tile_mask_t[i >> 4], 1u << k);  // OR of distinct bits: order-independent
}

// One wave per (site chunk, offset): partial dW_k over the chunk's sites, full CIN x COUT block in registers.
//   v_mfma_f32_16x16x4_f32:  D[row 16][col 16] += A[row 16][site 4] * B[site 4][col 16]
//   A lane (i, kq) = x[nbr[k][j0 + kq]][VA i .. VA i + VA-1]   -- ONE 16 / 8 / 4-byte load for the VA = CIN / 16 row blocks: row i
//                    of block cb is channel VA i + cb (the channel order inside dW is ours to choose; it is undone by the store)
//   B lane (n, kq) = dy[j0 + kq][VB n .. VB n + VB-1]           likewise, VB = COUT / 16
//   (missing neighbour / site beyond the live count / channel >= CIN: out-of-range buffer offset -> 0)
// 16-site tiles whose mask has no bit k are skipped wave-uniformly. The loop is software-pipelined over LIVE tiles: the operands
// of the next live tile (4 steps: 2 x 4 vector loads) are gathered before the MFMAs of the current one, and the neighbour
// indices of the tile after that before those gathers -- the first version issued index load -> gather -> MFMA per 4-site step
// with ~1.7 waves per SIMD to hide it behind: 96 us for a 64 -> 64 level at batch 4, almost all of it load latency. Every load
// is unconditional (a tile past the chunk's end gets out-of-range offsets), so the vmcnt waits are exact.
template <int N> struct VecLoad;
template <> struct VecLoad<1> {
  static __device__ __forceinline__ void ld(rsrc_t r, unsigned off, float* v) {
    v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
  }
};
template <> struct VecLoad<2> {
  static __device__ __forceinline__ void ld(rsrc_t r, unsigned off, float* v) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 t = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
    v[0] = t.x; v[1] = t.y;
  }
};
template <> struct VecLoad<4> {
  static __device__ __forceinline__ void ld(rsrc_t r, unsigned off, float* v) {
    const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};

template <int CIN, int COUT>
__global__ __launch_bounds__(64, 2) void wgrad_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const int* __restrict__ nbr,
                                                            const uint32_t* __restrict__ tile_mask,
                                                            const int* __restrict__ n_dev, int n_cap,
                                                            float* __restrict__ partial) {
  constexpr int VA = CIN >= 16 ? CIN / 16 : 1, VB = COUT / 16;
  static_assert(VA == 1 || VA == 2 || VA == 4, "CIN in {4, 16, 32, 64}");
  static_assert(VB == 1 || VB == 2 || VB == 4, "COUT in {16, 32, 64}");
  const int chunk = blockIdx.x, k = blockIdx.y, kv = gridDim.y;
  const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
  const int n = min(n_dev[0], n_cap);
  // the WG_CHUNKS chunks divide the LIVE tiles (device count), not the table's capacity: with capacity-sized tables (a captured
  // training iteration) chunks cut by capacity left the tail chunks empty and the others proportionally longer (measured: 231 us
  // instead of 98 for the 64 -> 64 layers), and the grouping of the partial sums -- hence the bits -- depended on the capacity
  const int live_tiles = (n + 15) >> 4;
  const int chunk_tiles = (live_tiles + WG_CHUNKS - 1) / WG_CHUNKS;
  const int t0 = chunk * chunk_tiles, t1 = min(t0 + chunk_tiles, live_tiles);
  f32x4 acc[VA][VB];
#pragma unroll
  for (int a = 0; a < VA; ++a)
#pragma unroll
    for (int b = 0; b < VB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const rsrc_t xr = make_rsrc(x, 0x7FFFFFFFu);
  const rsrc_t yr = make_rsrc(dy, (unsigned)min((long long)n * COUT * 4, 0x7FFFFFFFll));  // rows >= n read as 0
  const rsrc_t nr = make_rsrc(nbr + (size_t)k * n_cap, (unsigned)n_cap * 4u);
  const bool a_ok = CIN >= 16 || i < CIN;

  // first live tile at or after t (t1 if none): wave-uniform scalar walk over the mask words
#define SESSD_SW_NEXT(T)                                                  \
  {                                                                       \
    while ((T) < t1 && !((tile_mask[(T)] >> k) & 1u)) ++(T);              \
  }
  // the four neighbour indices of this lane in tile T (sites 4 s + kq); a tile past the end reads out of range (= 0, unused)
#define SESSD_SW_INDEX(R, T)                                                                                           \
  _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                         \
    (R)[s] = __builtin_amdgcn_raw_buffer_load_b32(nr, (T) < t1 ? (int)(((T) * 16 + s * 4 + kq) * 4) : (int)SESSD_OOB, 0, 0);
  // operands of tile T from its indices
#define SESSD_SW_GATHER(A, B, R, T)                                                                                    \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                       \
    const int j = (T) * 16 + s * 4 + kq;                                                                                \
    const int r = ((T) < t1 && j < n) ? (int)(R)[s] : -1;                                                               \
    VecLoad<VA>::ld(xr, (r >= 0 && a_ok) ? (unsigned)((r * CIN + VA * i) * 4) : SESSD_OOB, (A)[s]);                     \
    VecLoad<VB>::ld(yr, (T) < t1 ? (unsigned)((j * COUT + VB * i) * 4) : SESSD_OOB, (B)[s]);                            \
  }
#define SESSD_SW_MMA(A, B)                                                                                             \
  _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                         \
    _Pragma("unroll") for (int cb = 0; cb < VA; ++cb)                                                                   \
      _Pragma("unroll") for (int ob = 0; ob < VB; ++ob)                                                                 \
        acc[cb][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32((A)[s][cb], (B)[s][ob], acc[cb][ob], 0, 0, 0);

  int ta = t0;
  SESSD_SW_NEXT(ta)
  if (ta < t1) {
    unsigned ra[4], rb[4];
    float a0[4][VA], b0[4][VB], a1[4][VA], b1[4][VB];
    SESSD_SW_INDEX(ra, ta)
    int tb = ta + 1;
    SESSD_SW_NEXT(tb)
    SESSD_SW_INDEX(rb, tb)
    __builtin_amdgcn_sched_barrier(0);
    SESSD_SW_GATHER(a0, b0, ra, ta)
    __builtin_amdgcn_sched_barrier(0);
    while (ta < t1) {   // set 0 holds live tile ta, rb the indices of tb (past the end: everything about it reads as zero)
      SESSD_SW_GATHER(a1, b1, rb, tb)
      int tc = tb < t1 ? tb + 1 : t1;
      SESSD_SW_NEXT(tc)
      SESSD_SW_INDEX(ra, tc)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_SW_MMA(a0, b0)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_SW_GATHER(a0, b0, ra, tc)
      int td = tc < t1 ? tc + 1 : t1;
      SESSD_SW_NEXT(td)
      SESSD_SW_INDEX(rb, td)
      __builtin_amdgcn_sched_barrier(0);
      SESSD_SW_MMA(a1, b1)   // a tile past the end multiplies zeros
      __builtin_amdgcn_sched_barrier(0);
      ta = tc;
      tb = td;
    }
  }
#undef SESSD_SW_NEXT
#undef SESSD_SW_INDEX
#undef SESSD_SW_GATHER
#undef SESSD_SW_MMA
  // D layout: column = lane & 15, rows = (lane >> 4) * 4 + r; row rho of block cb is channel VA rho + cb, column n of block ob is
  // channel VB n + ob: the VB blocks of a lane are VB consecutive output channels
  float* dst = partial + ((size_t)chunk * kv + k) * CIN * COUT;
#pragma unroll
  for (int cb = 0; cb < VA; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = VA * (kq * 4 + r) + cb;
      if (ci < CIN) {
        float* o = dst + ci * COUT + VB * i;
#pragma unroll
        for (int ob = 0; ob < VB; ++ob) o[ob] = acc[cb][ob][r];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nchunks, int total,
                                                            float* __restrict__ grad_weight) {
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
  grad_weight[e] = s;
}

template <int CIN, int COUT>
int launch_wgrad(const float* x, const float* dy, const int* nbr, const uint32_t* tile_mask, int kv, const int* n_dev,
                 int n_cap, float* grad_weight, float* partial, hipStream_t stream) {
  const int nchunks = WG_CHUNKS;
  SESSD_LAUNCH((wgrad_partial_kernel<CIN, COUT>), dim3(nchunks, kv), dim3(64), 0, stream, x, dy, nbr, tile_mask,
                     n_dev, n_cap, partial);
  SESSD_CHECK_LAUNCH();
  const int total = kv * CIN * COUT;
  SESSD_LAUNCH(wgrad_reduce_kernel, dim3(sessd_divup(total, 256)), dim3(256), 0, stream, partial, nchunks, total,
                     grad_weight);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // namespace

extern "C" {

// nbr (kv, n_out_cap) of a conv -> nbr_t (kv, n_in_cap) with nbr_t[k][i] = j <=> nbr[k][j] = i (else -1) and the
// per-16-input-site offset masks: the rulebook of the data-gradient pass (dx = conv of dy over the INPUT sites).
int sessd_sparse_rulebook_transpose(const int* nbr, int kernel_volume, const int* n_out_dev, int n_out_cap,
                                    int n_in_cap, int* nbr_t, uint32_t* tile_mask_t, hipStream_t stream) {
  if (kernel_volume <= 0 || kernel_volume > 32 || n_out_cap <= 0 || n_in_cap <= 0) return SESSD_EINVAL;
  SESSD_FILL(nbr_t, 0xFFFFFFFFu, (size_t)kernel_volume * n_in_cap, stream);
  SESSD_FILL(tile_mask_t, 0u, (size_t)sessd_divup(n_in_cap, 16), stream);
  SESSD_LAUNCH(rulebook_transpose_kernel, dim3(sessd_divup(n_out_cap, 256), kernel_volume), dim3(256), 0, stream,
                     nbr, n_out_dev, n_out_cap, n_in_cap, nbr_t, tile_mask_t);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

size_t sessd_sparse_conv_wgrad_workspace_bytes(int kernel_volume, int cin, int cout) {
  return (size_t)WG_CHUNKS * kernel_volume * cin * cout * sizeof(float);
}

// grad_weight (kv, cin, cout) [= the spconv (kz,ky,kx,Cin,Cout) layout flattened] = sum over rulebook pairs of
// in_feat[nbr[k][j]] (x) grad_out[j]; in_feat (n_in, cin), grad_out (n_out_cap, cout) row-major float32.
int sessd_sparse_conv_wgrad(const float* in_feat, int cin, const float* grad_out, int cout, const int* nbr,
                            const uint32_t* tile_mask, int kernel_volume, const int* n_out_dev, int n_out_cap,
                            float* grad_weight, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (kernel_volume <= 0 || kernel_volume > 32 || n_out_cap <= 0) return SESSD_EINVAL;
  if (workspace_bytes < sessd_sparse_conv_wgrad_workspace_bytes(kernel_volume, cin, cout)) return SESSD_EWORKSPACE;
#define SESSD_WG(CI, CO)                                                                                       \
  if (cin == CI && cout == CO)                                                                                 \
    return launch_wgrad<CI, CO>(in_feat, grad_out, nbr, tile_mask, kernel_volume, n_out_dev, n_out_cap, grad_weight, \
                                (float*)workspace, stream);
  SESSD_WG(4, 16)
  SESSD_WG(16, 16)
  SESSD_WG(16, 32)
  SESSD_WG(32, 32)
  SESSD_WG(32, 64)
  SESSD_WG(64, 64)
#undef SESSD_WG
  return SESSD_EINVAL;  // channel pair not instantiated
}

}  // extern "C"
