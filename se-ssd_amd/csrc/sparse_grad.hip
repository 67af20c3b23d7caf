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
//   v_mfma_f32_16x16x4_f32:  D[ci 16][co 16] += A[ci 16][site 4] * B[site 4][co 16]
//   A lane (i, kq) = x[nbr[k][j0 + kq]][cb*16 + i]   (missing neighbour / channel >= CIN: out-of-range offset -> 0)
//   B lane (n, kq) = dy[j0 + kq][ob*16 + n]
// 16-site tiles whose mask has no bit k are skipped wave-uniformly.
template <int CIN, int COUT>
__global__ __launch_bounds__(64) void wgrad_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const int* __restrict__ nbr,
                                                            const uint32_t* __restrict__ tile_mask,
                                                            const int* __restrict__ n_dev, int n_cap,
                                                            float* __restrict__ partial) {
  constexpr int CIB = (CIN + 15) / 16, COB = COUT / 16;
  const int chunk = blockIdx.x, k = blockIdx.y, kv = gridDim.y;
  const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
  const int n = min(n_dev[0], n_cap);
  // the WG_CHUNKS chunks divide the LIVE tiles (device count), not the table's capacity: with capacity-sized tables (a captured
  // training iteration) chunks cut by capacity left the tail chunks empty and the others proportionally longer (measured: 231 us
  // instead of 98 for the 64 -> 64 layers), and the grouping of the partial sums -- hence the bits -- depended on the capacity
  const int live_tiles = (n + 15) >> 4;
  const int chunk_tiles = (live_tiles + WG_CHUNKS - 1) / WG_CHUNKS;
  const int t0 = chunk * chunk_tiles, t1 = min(t0 + chunk_tiles, live_tiles);
  f32x4 acc[CIB][COB];
#pragma unroll
  for (int a = 0; a < CIB; ++a)
#pragma unroll
    for (int b = 0; b < COB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const rsrc_t xr = make_rsrc(x, 0x7FFFFFFFu);
  const rsrc_t yr = make_rsrc(dy, (unsigned)min((long long)n * COUT * 4, 0x7FFFFFFFll));  // rows >= n read as 0
  const int* nb = nbr + (size_t)k * n_cap;
  for (int t = t0; t < t1; ++t) {
    if (!((tile_mask[t] >> k) & 1u)) continue;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = t * 16 + s * 4 + kq;
      const int r = j < n ? nb[j] : -1;
      float a[CIB], b[COB];
#pragma unroll
      for (int cb = 0; cb < CIB; ++cb)
        a[cb] = bufload1(xr, (r >= 0 && cb * 16 + i < CIN) ? (unsigned)((r * CIN + cb * 16 + i) * 4) : SESSD_OOB, 0);
#pragma unroll
      for (int ob = 0; ob < COB; ++ob) b[ob] = bufload1(yr, (unsigned)((j * COUT + ob * 16 + i) * 4), 0);
#pragma unroll
      for (int cb = 0; cb < CIB; ++cb)
#pragma unroll
        for (int ob = 0; ob < COB; ++ob)
          acc[cb][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb], b[ob], acc[cb][ob], 0, 0, 0);
    }
  }
  // D layout: column (co) = lane & 15, rows (ci) = (lane >> 4) * 4 + r
  float* dst = partial + ((size_t)chunk * kv + k) * CIN * COUT;
#pragma unroll
  for (int cb = 0; cb < CIB; ++cb)
#pragma unroll
    for (int ob = 0; ob < COB; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cb * 16 + kq * 4 + r;
        if (ci < CIN) dst[ci * COUT + ob * 16 + i] = acc[cb][ob][r];
      }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nchunks, int total,
                                                            float* __restrict__ grad_weight) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += partial[(size_t)c * total + e];
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
