// Train-mode BatchNorm1d + ReLU over a sparse level's feature table (N rows x C channels), forward and backward
// (DESIGN.md section 9 item 2). In the SE-SSD training step (det3d/torchie/trainer/trainer_sessd.py:250-275) both networks run
// SpMiddleFHD in train mode (scn.py:103-148: BatchNorm1d(eps=1e-3, momentum=0.01) + ReLU after each of the 14 sparse convs), which
// as torch modules costs ~5 launches forward and ~6 backward per layer on tables of 3 k - 60 k rows: launch-bound. Here each
// pass is TWO launches:
//   forward   statistics: per-block partial sums (float64) over a row chunk; the block that finishes LAST adds the partials in
//             block order and writes mean, 1/sqrt(var + eps) and the running statistics  ->  y = max(0, xhat * gamma + beta)
//   backward  dz = dy * [y > 0];  the same two-level sum of dz and dz * xhat -> dgamma, dbeta  ->
//             dx = gamma * invstd * (dz - dbeta / N - xhat * dgamma / N)
// "Last block" = the one whose increment of a counter word brings it to the grid size (partials stored through to the coherent
// level and acknowledged, then one atomic per block): no workgroup waits for another, and the order of the sum is
// fixed, so the result does not depend on which block arrives last. (The first version used the textbook __threadfence()
// pair; on this chip that is a write-back + invalidate of the block's whole L2, paid by every block.) The counter lives in the first 256 bytes of the workspace:
// ZERO on entry (the caller clears a new workspace once), left zero by every call. The row count N stays on the device; rows
// >= N of y / dx are written as zeros (the capacity form of the tables, spconv/__init__.py). HBM-bound: 2 passes over x forward,
// 2 over (x, dy, y) backward, 16-byte accesses.
//
// Second half of the file: the same for the dense BEV layout (B, C, H, W) of the SSFA neck in train mode
// (det3d/models/necks/rpn_v1.py:131-210: BatchNorm2d(eps=1e-3, momentum=0.01) + ReLU after each of its 13 convolutions) --
// one block per (channel, plane slice), 16-byte accesses along the plane; the last slice of a CHANNEL to finish finalises that
// channel (one counter word per channel).
// tests/test_bn_train_gpu.py compares both with torch.nn.BatchNorm1d / BatchNorm2d + ReLU (forward, running statistics, gradients).
#include "common.hpp"

namespace {

constexpr int BN_BLOCKS = 128;        // statistics blocks. The finishing block reads 2 x C doubles per block past the L2 (~65 GB/s for
                                      // one block): 256 blocks made the finisher the longer half of the launch (14.4 us), 64 starve the
                                      // row pass; measured on the captured iteration: 256 -> 15.09 ms, 128 -> 15.01, 64 -> 15.16
constexpr int NT = 256;
constexpr size_t BN_COUNTER_BYTES = 256;

struct BnFinal {                      // what the finishing block writes
  float eps, momentum;
  float* running_mean;                // forward (both or neither)
  float* running_var;
  float* save_mean;
  float* save_invstd;
  float* dgamma;                      // backward
  float* dbeta;
  double* sums;                       // SyncBN form (not NULL): the finishing block ALSO / INSTEAD writes the raw totals here --
                                      // forward: [sum x (C), sum x^2 (C), rows]; backward: [sum dz (C), sum dz * xhat (C)] -- to be
                                      // all-reduced over the ranks before the apply launch (det3d/torchie/apis/train_sessd.py:286-294)
};

// Partial sums travel between workgroups (possibly on different XCDs, each with its own L2) as agent-scope relaxed atomic
// stores / loads of the individual words: they go through to / come from the coherent level without the whole-L2 write-back
// and invalidate that a __threadfence() pair costs every block (measured: the fence form made the iteration 10 % slower).
// Order: a block's stores are acknowledged (vmcnt 0) before its thread 0 counts the block in.
__device__ __forceinline__ void put_partial(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double get_partial(const double* p) {
  return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define SESSD_STORES_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// Two per-channel sums over the rows of this block's chunk; thread = (row lane, group of VEC channels); C a power of two <= 256,
// VEC = 4 when C % 4 == 0 (one 16-byte load per row and thread), else 1.
template <bool BWD, int VEC>
__global__ __launch_bounds__(NT) void bn_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const int* __restrict__ n_dev, int n_cap,
                                                       int C, int relu, double* partial, unsigned* counter, BnFinal F) {
  __shared__ double sm[2 * VEC][NT];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const int n = min(n_dev[0], n_cap);
  const int Q = C / VEC;                    // channel groups per row
  const int lanes = NT / Q;                 // rows per block and step
  const int q = tid % Q, rl = tid / Q;
  const int chunk = sessd_divup(n, (int)gridDim.x);
  const int r0 = blockIdx.x * chunk, r1 = min(n, r0 + chunk);
  double s0[VEC], s1[VEC];
  float mu[VEC], is[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    s0[v] = 0.0; s1[v] = 0.0;
    mu[v] = BWD ? mean[q * VEC + v] : 0.f;
    is[v] = BWD ? invstd[q * VEC + v] : 0.f;
  }
#pragma unroll 4
  for (int r = r0 + rl; r < r1; r += lanes) {
    const size_t o = (size_t)r * C + q * VEC;
    float xv[VEC], dz[VEC], yv[VEC];
    if (VEC == 4) {
      *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + o);
      if (BWD) {
        *reinterpret_cast<float4*>(dz) = *reinterpret_cast<const float4*>(dy + o);
        if (relu) *reinterpret_cast<float4*>(yv) = *reinterpret_cast<const float4*>(y + o);
      }
    } else {
      xv[0] = x[o];
      if (BWD) { dz[0] = dy[o]; if (relu) yv[0] = y[o]; }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (!BWD) {
        const double d = xv[v];
        s0[v] += d;
        s1[v] += d * d;
      } else {
        float g = dz[v];
        if (relu && !(yv[v] > 0.f)) g = 0.f;
        s0[v] += (double)g;
        s1[v] += (double)g * (double)((xv[v] - mu[v]) * is[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) { sm[v][tid] = s0[v]; sm[VEC + v][tid] = s1[v]; }
  __syncthreads();
  for (int s = NT / 2; s >= Q; s >>= 1) {   // tid + s has the same channel group (Q, s powers of two, s >= Q)
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 2 * VEC; ++k) sm[k][tid] += sm[k][tid + s];
    }
    __syncthreads();
  }
  if (tid < Q) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      put_partial(partial + ((size_t)blockIdx.x * 2 + 0) * C + tid * VEC + v, sm[v][tid]);
      put_partial(partial + ((size_t)blockIdx.x * 2 + 1) * C + tid * VEC + v, sm[VEC + v][tid]);
    }
  }
  // ---- the last block to arrive adds the partials of all blocks, in block order
  SESSD_STORES_DONE();
  __syncthreads();
  if (tid == 0) s_last = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  const int G = NT / C, c = tid % C, g = tid / C;   // G groups of blocks per channel
  double t0 = 0.0, t1 = 0.0;
  const int nb = (int)gridDim.x;
  for (int b = g; b < nb; b += 16 * G) {   // 32 loads in flight per thread: these loads go past the L2, a round trip each
    double a0[16], a1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int bb = b + k * G;
      const bool ok = bb < nb;
      a0[k] = ok ? get_partial(partial + ((size_t)bb * 2 + 0) * C + c) : 0.0;
      a1[k] = ok ? get_partial(partial + ((size_t)bb * 2 + 1) * C + c) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { t0 += a0[k]; t1 += a1[k]; }
  }
  sm[0][tid] = t0;
  sm[1][tid] = t1;
  __syncthreads();
  for (int s = NT / 2; s >= C; s >>= 1) {
    if (tid < s) { sm[0][tid] += sm[0][tid + s]; sm[1][tid] += sm[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid >= C) return;
  t0 = sm[0][tid];
  t1 = sm[1][tid];
  if (F.sums) {   // SyncBN: the totals of THIS rank; mean / invstd / running statistics follow the all-reduce (bn_sync_finalize)
    F.sums[c] = t0;
    F.sums[C + c] = t1;
    if (!BWD && c == 0) F.sums[2 * C] = (double)n;
    if (BWD) { F.dbeta[c] = (float)t0; F.dgamma[c] = (float)t1; }   // the parameter gradients stay local (averaged with all others)
    return;
  }
  if (!BWD) {
    const double m = n > 0 ? t0 / n : 0.0;
    double var = n > 0 ? t1 / n - m * m : 0.0;
    if (var < 0.0) var = 0.0;
    F.save_mean[c] = (float)m;
    F.save_invstd[c] = (float)(1.0 / sqrt(var + (double)F.eps));
    if (F.running_mean && n > 0) {   // torch: running = (1 - momentum) * running + momentum * batch, variance unbiased
      const double unbiased = n > 1 ? var * n / (n - 1) : var;
      F.running_mean[c] = (float)((1.0 - F.momentum) * F.running_mean[c] + F.momentum * m);
      F.running_var[c] = (float)((1.0 - F.momentum) * F.running_var[c] + F.momentum * unbiased);
    }
  } else {
    F.dbeta[c] = (float)t0;
    F.dgamma[c] = (float)t1;
  }
}

// thread = VEC consecutive channels of one row; rows >= *n_dev (up to n_cap) are written as zeros
template <bool BWD, int VEC>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ y_in, const int* __restrict__ n_dev, int n_cap,
                                                       int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu,
                                                       float* __restrict__ out, const float* __restrict__ inv_n_sync) {
  const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * VEC;
  if (i >= (size_t)n_cap * C) return;
  const int n = min(n_dev[0], n_cap);
  const int c0 = (int)(i & (size_t)(C - 1));   // C is a power of two
  float r[VEC];
  if (i >= (size_t)n * C) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) r[v] = 0.f;
  } else {
    float xv[VEC], dz[VEC], yv[VEC];
    if (VEC == 4) {
      *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + i);
      if (BWD) {
        *reinterpret_cast<float4*>(dz) = *reinterpret_cast<const float4*>(dy + i);
        if (relu) *reinterpret_cast<float4*>(yv) = *reinterpret_cast<const float4*>(y_in + i);
      }
    } else {
      xv[0] = x[i];
      if (BWD) { dz[0] = dy[i]; if (relu) yv[0] = y_in[i]; }
    }
    const float inv_n = inv_n_sync ? inv_n_sync[0] : 1.f / (float)n;   // SyncBN: 1 / rows of ALL ranks
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int c = c0 + v;
      const float g = gamma ? gamma[c] : 1.f;
      if (!BWD) {
        float t = (xv[v] - mean[c]) * invstd[c] * g + (beta ? beta[c] : 0.f);
        if (relu) t = fmaxf(t, 0.f);
        r[v] = t;
      } else {
        float d = dz[v];
        if (relu && !(yv[v] > 0.f)) d = 0.f;
        const float xhat = (xv[v] - mean[c]) * invstd[c];
        r[v] = g * invstd[c] * (d - dbeta[c] * inv_n - xhat * dgamma[c] * inv_n);
      }
    }
  }
  if (VEC == 4)
    *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(r);
  else
    out[i] = r[0];
}

bool channels_ok(int C) { return C >= 1 && C <= NT && (C & (C - 1)) == 0; }

}  // namespace

extern "C" {

// first BN_COUNTER_BYTES: the arrival counter (zero on entry, zero on return), then the per-block partial sums
size_t sessd_bn_relu_train_workspace_bytes(int channels) {
  return BN_COUNTER_BYTES + (size_t)BN_BLOCKS * 2 * channels * sizeof(double);
}

// y = relu?( (x - mean) * invstd * gamma + beta ) with the batch statistics of rows < *n_dev of x (n_cap, channels);
// save_mean / save_invstd (channels) for the backward; running_mean / running_var updated in place when not NULL
// (momentum, unbiased variance: torch.nn.BatchNorm1d semantics). channels: a power of two <= 256. The workspace's first 256
// bytes must be zero on entry (they are zero again on return).
int sessd_bn_relu_train_fwd(const float* x, const int* n_dev, int n_cap, int channels, const float* gamma, const float* beta,
                            float eps, float momentum, int relu, float* running_mean, float* running_var, float* y,
                            float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels)) return SESSD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counter = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN_COUNTER_BYTES);
  BnFinal F{eps, momentum, running_mean, running_var, save_mean, save_invstd, nullptr, nullptr, nullptr};
  const size_t total = (size_t)n_cap * channels;
  const float* nf = nullptr;
  if (channels % 4 == 0) {
    SESSD_LAUNCH((bn_stats_kernel<false, 4>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, nf, nf, nf, nf, n_dev, n_cap, channels, 0,
                 partial, counter, F);
    SESSD_CHECK_LAUNCH();
    SESSD_LAUNCH((bn_apply_kernel<false, 4>), dim3((unsigned)((total / 4 + NT - 1) / NT)), dim3(NT), 0, stream, x, nf, nf, n_dev,
                 n_cap, channels, gamma, beta, (const float*)save_mean, (const float*)save_invstd, nf, nf, relu, y, nf);
  } else {
    SESSD_LAUNCH((bn_stats_kernel<false, 1>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, nf, nf, nf, nf, n_dev, n_cap, channels, 0,
                 partial, counter, F);
    SESSD_CHECK_LAUNCH();
    SESSD_LAUNCH((bn_apply_kernel<false, 1>), dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, stream, x, nf, nf, n_dev, n_cap,
                 channels, gamma, beta, (const float*)save_mean, (const float*)save_invstd, nf, nf, relu, y, nf);
  }
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// gradients of the above: dx (n_cap, channels; rows >= *n_dev zero), dgamma, dbeta (channels). y is the forward output
// (the ReLU mask), x the forward input. Same workspace contract.
int sessd_bn_relu_train_bwd(const float* dy, const float* x, const float* y, const int* n_dev, int n_cap, int channels,
                            const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dx,
                            float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels) || !dgamma || !dbeta) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counter = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN_COUNTER_BYTES);
  BnFinal F{0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dgamma, dbeta, nullptr};
  const size_t total = (size_t)n_cap * channels;
  const float* nf = nullptr;
  if (channels % 4 == 0) {
    SESSD_LAUNCH((bn_stats_kernel<true, 4>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, n_dev, n_cap,
                 channels, relu, partial, counter, F);
    SESSD_CHECK_LAUNCH();
    SESSD_LAUNCH((bn_apply_kernel<true, 4>), dim3((unsigned)((total / 4 + NT - 1) / NT)), dim3(NT), 0, stream, x, dy, y, n_dev,
                 n_cap, channels, gamma, nf, save_mean, save_invstd, (const float*)dgamma, (const float*)dbeta, relu, dx, nf);
  } else {
    SESSD_LAUNCH((bn_stats_kernel<true, 1>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, n_dev, n_cap,
                 channels, relu, partial, counter, F);
    SESSD_CHECK_LAUNCH();
    SESSD_LAUNCH((bn_apply_kernel<true, 1>), dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, stream, x, dy, y, n_dev, n_cap,
                 channels, gamma, nf, save_mean, save_invstd, (const float*)dgamma, (const float*)dbeta, relu, dx, nf);
  }
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ dense (B, C, H, W) layout
namespace {

constexpr int BN2D_SPLIT = 16;   // plane slices per channel: C x 16 blocks of partial sums

// The normalised, scaled and shifted value before the ReLU. This file is compiled with -ffp-contract=off (build.py), so the
// expression is the same four roundings (sub, mul, mul, add) wherever it is inlined: the forward writes max(0, z) and a backward
// that is not given y re-derives the ReLU mask as z > 0 from x -- the same bits, so the same mask
// (tests/test_bn_train_gpu.py::test_dense_backward_mask_from_x_equals_mask_from_y). The FORM matters beyond that: with an fma in
// its place (one rounding fewer) the whole-detector gradients moved by 1.5e-3 in norm on the layers behind deconv_block_0 --
// the BEV map is mostly empty, thousands of pixels of a channel carry the SAME value, and where that value sits at the ReLU
// threshold a last-bit change switches all of them at once; the four-rounding form is the one torch's CPU BatchNorm (the
// oracle) agrees with to 1e-5 (scripts/dbg_grad_dump.py, three hybrid builds).
__device__ __forceinline__ float bn2d_z(float x, float mu, float is, float g, float bt) { return (x - mu) * is * g + bt; }

constexpr int BN2D_MAX_CHANNELS = 1024;
// one counter word per channel in a region of FIXED size: one workspace serves calls with different channel counts
constexpr size_t BN2D_COUNTER_BYTES = (size_t)BN2D_MAX_CHANNELS * 4;

// block (c, s): the two sums over the pixels [s * chunk, (s + 1) * chunk) of channel c in every image; plane % 4 == 0.
// MODE 0: sums of x and x^2 -> mean / invstd / running statistics; 1: sums of dz and dz * xhat -> dbeta / dgamma;
// 2: sum of x -> dbeta (a conv's bias gradient). The slice of a channel that finishes last adds the channel's 16 partials in
// slice order and writes the result (counter word per channel, zero on entry and on return).
template <int MODE>
__global__ __launch_bounds__(NT) void bn2d_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int B, int C, int plane, int relu,
                                                         double* partial, unsigned* counters, BnFinal F) {
  // MODE 1 with y == NULL: the ReLU mask is re-derived from x (gamma / beta of the forward; one tensor less to read)
  constexpr bool BWD = MODE == 1;
  __shared__ double sm[2][NT / 64];
  const int c = blockIdx.x, s = blockIdx.y;
  const int quads = plane >> 2;
  const int chunk = sessd_divup(quads, BN2D_SPLIT);
  const int q0 = s * chunk, q1 = min(quads, q0 + chunk);
  double s0 = 0.0, s1 = 0.0;
  float mu = 0.f, is = 0.f, gm = 1.f, bt = 0.f;
  if (BWD) {
    mu = mean[c]; is = invstd[c];
    if (gamma) gm = gamma[c];
    if (beta) bt = beta[c];
  }
  for (int b = 0; b < B; ++b) {
    const size_t base = ((size_t)b * C + c) * plane;
    for (int q = q0 + (int)threadIdx.x; q < q1; q += NT) {
      const float4 xv = *reinterpret_cast<const float4*>(x + base + 4 * (size_t)q);
      if (!BWD) {
        s0 += (double)xv.x + (double)xv.y + (double)xv.z + (double)xv.w;
        if (MODE == 0) s1 += (double)xv.x * xv.x + (double)xv.y * xv.y + (double)xv.z * xv.z + (double)xv.w * xv.w;
      } else {
        float4 dz = *reinterpret_cast<const float4*>(dy + base + 4 * (size_t)q);
        if (relu) {
          float4 yv;
          if (y) {
            yv = *reinterpret_cast<const float4*>(y + base + 4 * (size_t)q);
          } else {
            yv.x = bn2d_z(xv.x, mu, is, gm, bt); yv.y = bn2d_z(xv.y, mu, is, gm, bt);
            yv.z = bn2d_z(xv.z, mu, is, gm, bt); yv.w = bn2d_z(xv.w, mu, is, gm, bt);
          }
          if (!(yv.x > 0.f)) dz.x = 0.f;
          if (!(yv.y > 0.f)) dz.y = 0.f;
          if (!(yv.z > 0.f)) dz.z = 0.f;
          if (!(yv.w > 0.f)) dz.w = 0.f;
        }
        s0 += (double)dz.x + (double)dz.y + (double)dz.z + (double)dz.w;
        s1 += (double)dz.x * (double)((xv.x - mu) * is) + (double)dz.y * (double)((xv.y - mu) * is) +
              (double)dz.z * (double)((xv.z - mu) * is) + (double)dz.w * (double)((xv.w - mu) * is);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s0; sm[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double t0 = 0.0, t1 = 0.0;
  for (int w = 0; w < NT / 64; ++w) { t0 += sm[0][w]; t1 += sm[1][w]; }
  double* pc = partial + (size_t)c * BN2D_SPLIT * 2;
  put_partial(pc + s * 2 + 0, t0);
  put_partial(pc + s * 2 + 1, t1);
  SESSD_STORES_DONE();
  if (__hip_atomic_fetch_add(counters + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != BN2D_SPLIT - 1) return;
  double a0[BN2D_SPLIT], a1[BN2D_SPLIT];
#pragma unroll
  for (int k = 0; k < BN2D_SPLIT; ++k) { a0[k] = get_partial(pc + k * 2 + 0); a1[k] = get_partial(pc + k * 2 + 1); }
  t0 = 0.0; t1 = 0.0;
#pragma unroll
  for (int k = 0; k < BN2D_SPLIT; ++k) { t0 += a0[k]; t1 += a1[k]; }
  __hip_atomic_store(counters + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (F.sums && MODE != 2) {   // SyncBN: this rank's totals (see BnFinal); the rest follows the all-reduce
    F.sums[c] = t0;
    F.sums[C + c] = t1;
    if (MODE == 0 && c == 0) F.sums[2 * C] = (double)B * (double)plane;
    if (MODE == 1) { F.dbeta[c] = (float)t0; F.dgamma[c] = (float)t1; }
    return;
  }
  if (MODE == 0) {
    const double n = (double)B * (double)plane;
    const double m = t0 / n;
    double var = t1 / n - m * m;
    if (var < 0.0) var = 0.0;
    F.save_mean[c] = (float)m;
    F.save_invstd[c] = (float)(1.0 / sqrt(var + (double)F.eps));
    if (F.running_mean) {
      const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
      F.running_mean[c] = (float)((1.0 - F.momentum) * F.running_mean[c] + F.momentum * m);
      F.running_var[c] = (float)((1.0 - F.momentum) * F.running_var[c] + F.momentum * unbiased);
    }
  } else {
    F.dbeta[c] = (float)t0;
    if (MODE == 1) F.dgamma[c] = (float)t1;
  }
}

// one thread = four consecutive pixels of one (image, channel) plane
template <bool BWD>
__global__ __launch_bounds__(NT) void bn2d_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ y_in, int C, int plane,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                         float inv_n, int relu, float* __restrict__ out, size_t total_quads,
                                                         const float* __restrict__ inv_n_sync) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= total_quads) return;
  if (inv_n_sync) inv_n = inv_n_sync[0];   // SyncBN: 1 / (pixels of ALL ranks)
  const int c = (int)((i / (size_t)(plane >> 2)) % (size_t)C);
  const float mu = mean[c], is = invstd[c], g = gamma ? gamma[c] : 1.f;
  const float4 xv = *reinterpret_cast<const float4*>(x + 4 * i);
  float4 r;
  if (!BWD) {
    const float bt = beta ? beta[c] : 0.f;
    r.x = bn2d_z(xv.x, mu, is, g, bt); r.y = bn2d_z(xv.y, mu, is, g, bt); r.z = bn2d_z(xv.z, mu, is, g, bt); r.w = bn2d_z(xv.w, mu, is, g, bt);
    if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
  } else {
    float4 dz = *reinterpret_cast<const float4*>(dy + 4 * i);
    if (relu) {
      float4 yv;
      if (y_in) {
        yv = *reinterpret_cast<const float4*>(y_in + 4 * i);
      } else {   // the mask from x (see bn2d_z)
        const float bt = beta ? beta[c] : 0.f;
        yv.x = bn2d_z(xv.x, mu, is, g, bt); yv.y = bn2d_z(xv.y, mu, is, g, bt);
        yv.z = bn2d_z(xv.z, mu, is, g, bt); yv.w = bn2d_z(xv.w, mu, is, g, bt);
      }
      if (!(yv.x > 0.f)) dz.x = 0.f;
      if (!(yv.y > 0.f)) dz.y = 0.f;
      if (!(yv.z > 0.f)) dz.z = 0.f;
      if (!(yv.w > 0.f)) dz.w = 0.f;
    }
    const float db = dbeta[c] * inv_n, dg = dgamma[c] * inv_n, k = g * is;
    r.x = k * (dz.x - db - (xv.x - mu) * is * dg); r.y = k * (dz.y - db - (xv.y - mu) * is * dg);
    r.z = k * (dz.z - db - (xv.z - mu) * is * dg); r.w = k * (dz.w - db - (xv.w - mu) * is * dg);
  }
  *reinterpret_cast<float4*>(out + 4 * i) = r;
}

}  // namespace

extern "C" {

// first: one arrival counter per channel (4 KB whatever the channel count; zero on entry, zero on return), then the slice
// partials. channels <= 1024.
size_t sessd_bn2d_relu_train_workspace_bytes(int channels) {
  return BN2D_COUNTER_BYTES + (size_t)channels * BN2D_SPLIT * 2 * sizeof(double);
}

// out[c] = sum over images and pixels of x[b][c][.] (the bias gradient of a conv: det3d's heads): the statistics pass of the
// train-mode BatchNorm above with only its first sum kept -- deterministic, no atomics, no memset (a torch reduction of this
// size clears a semaphore buffer with a memset, which a replayed hipGraph does not execute correctly on this stack).
int sessd_nchw_channel_sum(const float* x, int batch, int channels, int plane, float* out, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
  if (batch < 1 || channels < 1 || channels > BN2D_MAX_CHANNELS || plane < 4 || (plane & 3)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counters = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN2D_COUNTER_BYTES);
  BnFinal F{0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, out, nullptr};
  const float* nf = nullptr;
  SESSD_LAUNCH((bn2d_stats_kernel<2>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, nf, nf, nf, nf, nf, nf, batch, channels, plane, 0,
               partial, counters, F);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// BatchNorm2d (train mode) + optional ReLU on x (batch, channels, plane = H * W; plane % 4 == 0), torch.nn.BatchNorm2d semantics
// (biased batch variance to normalise, unbiased into running_var); save_mean / save_invstd (channels) for the backward.
int sessd_bn2d_relu_train_fwd(const float* x, int batch, int channels, int plane, const float* gamma, const float* beta, float eps,
                              float momentum, int relu, float* running_mean, float* running_var, float* y, float* save_mean,
                              float* save_invstd, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || channels > BN2D_MAX_CHANNELS || plane <= 0 || (plane & 3)) return SESSD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counters = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN2D_COUNTER_BYTES);
  BnFinal F{eps, momentum, running_mean, running_var, save_mean, save_invstd, nullptr, nullptr, nullptr};
  const float* nf = nullptr;
  SESSD_LAUNCH((bn2d_stats_kernel<0>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, nf, nf, nf, nf, nf, nf, batch, channels, plane, 0,
               partial, counters, F);
  SESSD_CHECK_LAUNCH();
  const size_t quads = (size_t)batch * channels * (plane >> 2);
  SESSD_LAUNCH((bn2d_apply_kernel<false>), dim3((unsigned)((quads + NT - 1) / NT)), dim3(NT), 0, stream, x, (const float*)nullptr,
               (const float*)nullptr, channels, plane, gamma, beta, save_mean, save_invstd, (const float*)nullptr,
               (const float*)nullptr, 0.f, relu, y, quads, (const float*)nullptr);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// gradients of the above: dx (same shape), dgamma, dbeta (channels); y = the forward output (ReLU mask), x = the forward input
static int bn2d_bwd_launch(const float* dy, const float* x, const float* y, int batch, int channels, int plane, const float* gamma,
                           const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma,
                           float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || channels > BN2D_MAX_CHANNELS || plane <= 0 || (plane & 3)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  if (!dgamma || !dbeta) return SESSD_EINVAL;
  unsigned* counters = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN2D_COUNTER_BYTES);
  BnFinal F{0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dgamma, dbeta, nullptr};
  SESSD_LAUNCH((bn2d_stats_kernel<1>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, gamma, beta,
               batch, channels, plane, relu, partial, counters, F);
  SESSD_CHECK_LAUNCH();
  const size_t quads = (size_t)batch * channels * (plane >> 2);
  SESSD_LAUNCH((bn2d_apply_kernel<true>), dim3((unsigned)((quads + NT - 1) / NT)), dim3(NT), 0, stream, x, dy, y, channels, plane,
               gamma, beta, save_mean, save_invstd, (const float*)dgamma, (const float*)dbeta,
               1.f / (float)((long long)batch * plane), relu, dx, quads, (const float*)nullptr);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_bn2d_relu_train_bwd(const float* dy, const float* x, const float* y, int batch, int channels, int plane,
                              const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dx,
                              float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (relu && !y) return SESSD_EINVAL;
  return bn2d_bwd_launch(dy, x, y, batch, channels, plane, gamma, nullptr, save_mean, save_invstd, relu, dx, dgamma, dbeta, workspace,
                         workspace_bytes, stream);
}

// The same without the forward output: the ReLU mask is re-derived from x with the forward's gamma / beta (the forward computes
// z with the same roundings, so z > 0 here is y > 0 there, bit for bit) -- one tensor less to read in both launches.
int sessd_bn2d_relu_train_bwd_x(const float* dy, const float* x, int batch, int channels, int plane, const float* gamma,
                                const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dx,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return bn2d_bwd_launch(dy, x, nullptr, batch, channels, plane, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta,
                         workspace, workspace_bytes, stream);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ SyncBN form (world size > 1)
// The reference's distributed path converts every BatchNorm to SyncBN (det3d/torchie/apis/train_sessd.py:286-294: apex
// convert_syncbn_model; the in-tree twin det3d/ops/syncbn/syncbn.py:37-103): batch statistics over the batches of ALL ranks.
// Each pass is split at the point where the ranks must talk:
//   *_stats      the statistics launch above; its finishing block writes this rank's raw totals (float64) instead of finalising
//   [host]       ONE all-reduce (sum) of those 2C + 1 / 2C doubles over RCCL
//   *_apply      bn_sync_finalize (one block: mean / invstd / running statistics, or the global gradient sums and 1 / N) + the
//                apply launch above
// With one rank the two forms give the same bits (same sums, same finalisation arithmetic).
namespace {

// forward: sums = [sum x (C), sum x^2 (C), N] over all ranks. backward: sums = [sum dz (C), sum dz * xhat (C)], n_total = fwd N.
template <bool BWD>
__global__ void bn_sync_finalize_kernel(const double* __restrict__ sums, const double* __restrict__ fwd_sums, int C, float eps,
                                        float momentum, float* running_mean, float* running_var, float* save_mean,
                                        float* save_invstd, float* g_dbeta, float* g_dgamma, float* inv_n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (!BWD) {
    const double n = sums[2 * C];
    const double m = n > 0 ? sums[c] / n : 0.0;
    double var = n > 0 ? sums[C + c] / n - m * m : 0.0;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)m;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && n > 0) {
      const double unbiased = n > 1 ? var * n / (n - 1) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  } else {
    g_dbeta[c] = (float)sums[c];
    g_dgamma[c] = (float)sums[C + c];
    if (c == 0) inv_n[0] = (float)(1.0 / fwd_sums[2 * C]);
  }
}

}  // namespace

extern "C" {

// scratch of the *_apply calls after the counters / partials: global dbeta, dgamma (C floats each) and 1 / N
size_t sessd_bn_sync_scratch_bytes(int channels) { return ((size_t)2 * channels + 4) * sizeof(float); }

int sessd_bn_relu_train_stats(const float* x, const int* n_dev, int n_cap, int channels, double* sums, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels) || !sums) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counter = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN_COUNTER_BYTES);
  BnFinal F{0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sums};
  const float* nf = nullptr;
  if (channels % 4 == 0)
    SESSD_LAUNCH((bn_stats_kernel<false, 4>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, nf, nf, nf, nf, n_dev, n_cap, channels, 0,
                 partial, counter, F);
  else
    SESSD_LAUNCH((bn_stats_kernel<false, 1>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, nf, nf, nf, nf, n_dev, n_cap, channels, 0,
                 partial, counter, F);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// sums: the all-reduced totals [sum x, sum x^2, N]
int sessd_bn_relu_train_apply(const float* x, const int* n_dev, int n_cap, int channels, const float* gamma, const float* beta,
                              float eps, float momentum, int relu, const double* sums, float* running_mean, float* running_var,
                              float* y, float* save_mean, float* save_invstd, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels) || !sums || !save_mean || !save_invstd) return SESSD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SESSD_EINVAL;
  SESSD_LAUNCH((bn_sync_finalize_kernel<false>), dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, sums, (const double*)nullptr,
               channels, eps, momentum, running_mean, running_var, save_mean, save_invstd, (float*)nullptr, (float*)nullptr,
               (float*)nullptr);
  SESSD_CHECK_LAUNCH();
  const size_t total = (size_t)n_cap * channels;
  const float* nf = nullptr;
  if (channels % 4 == 0)
    SESSD_LAUNCH((bn_apply_kernel<false, 4>), dim3((unsigned)((total / 4 + NT - 1) / NT)), dim3(NT), 0, stream, x, nf, nf, n_dev,
                 n_cap, channels, gamma, beta, (const float*)save_mean, (const float*)save_invstd, nf, nf, relu, y, nf);
  else
    SESSD_LAUNCH((bn_apply_kernel<false, 1>), dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, stream, x, nf, nf, n_dev, n_cap,
                 channels, gamma, beta, (const float*)save_mean, (const float*)save_invstd, nf, nf, relu, y, nf);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// this rank's sums [sum dz, sum dz * xhat] into `sums` (to be all-reduced) and into dgamma / dbeta (the parameter gradients: local)
int sessd_bn_relu_train_bwd_stats(const float* dy, const float* x, const float* y, const int* n_dev, int n_cap, int channels,
                                  const float* save_mean, const float* save_invstd, int relu, float* dgamma, float* dbeta,
                                  double* sums, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels) || !dgamma || !dbeta || !sums) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counter = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN_COUNTER_BYTES);
  BnFinal F{0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dgamma, dbeta, sums};
  if (channels % 4 == 0)
    SESSD_LAUNCH((bn_stats_kernel<true, 4>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, n_dev, n_cap,
                 channels, relu, partial, counter, F);
  else
    SESSD_LAUNCH((bn_stats_kernel<true, 1>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, n_dev, n_cap,
                 channels, relu, partial, counter, F);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// sums: all-reduced [sum dz, sum dz * xhat]; fwd_sums: the forward's all-reduced totals (its last entry is N of all ranks);
// scratch: sessd_bn_sync_scratch_bytes(channels) bytes
int sessd_bn_relu_train_bwd_apply(const float* dy, const float* x, const float* y, const int* n_dev, int n_cap, int channels,
                                  const float* gamma, const float* save_mean, const float* save_invstd, int relu, const double* sums,
                                  const double* fwd_sums, float* dx, void* scratch, size_t scratch_bytes, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels) || !sums || !fwd_sums || !scratch) return SESSD_EINVAL;
  if (scratch_bytes < sessd_bn_sync_scratch_bytes(channels)) return SESSD_EWORKSPACE;
  float* gdb = (float*)scratch;
  float* gdg = gdb + channels;
  float* inv_n = gdg + channels;
  SESSD_LAUNCH((bn_sync_finalize_kernel<true>), dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, sums, fwd_sums, channels, 0.f, 0.f,
               (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, gdb, gdg, inv_n);
  SESSD_CHECK_LAUNCH();
  const size_t total = (size_t)n_cap * channels;
  const float* nf = nullptr;
  if (channels % 4 == 0)
    SESSD_LAUNCH((bn_apply_kernel<true, 4>), dim3((unsigned)((total / 4 + NT - 1) / NT)), dim3(NT), 0, stream, x, dy, y, n_dev, n_cap,
                 channels, gamma, nf, save_mean, save_invstd, (const float*)gdg, (const float*)gdb, relu, dx, (const float*)inv_n);
  else
    SESSD_LAUNCH((bn_apply_kernel<true, 1>), dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, stream, x, dy, y, n_dev, n_cap,
                 channels, gamma, nf, save_mean, save_invstd, (const float*)gdg, (const float*)gdb, relu, dx, (const float*)inv_n);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// ---- the same four for the dense (batch, channels, plane) layout
int sessd_bn2d_relu_train_stats(const float* x, int batch, int channels, int plane, double* sums, void* workspace,
                                size_t workspace_bytes, hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || channels > BN2D_MAX_CHANNELS || plane <= 0 || (plane & 3) || !sums) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counters = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN2D_COUNTER_BYTES);
  BnFinal F{0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sums};
  const float* nf = nullptr;
  SESSD_LAUNCH((bn2d_stats_kernel<0>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, nf, nf, nf, nf, nf, nf, batch, channels, plane, 0,
               partial, counters, F);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_bn2d_relu_train_apply(const float* x, int batch, int channels, int plane, const float* gamma, const float* beta, float eps,
                                float momentum, int relu, const double* sums, float* running_mean, float* running_var, float* y,
                                float* save_mean, float* save_invstd, hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || channels > BN2D_MAX_CHANNELS || plane <= 0 || (plane & 3) || !sums || !save_mean || !save_invstd)
    return SESSD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SESSD_EINVAL;
  SESSD_LAUNCH((bn_sync_finalize_kernel<false>), dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, sums, (const double*)nullptr,
               channels, eps, momentum, running_mean, running_var, save_mean, save_invstd, (float*)nullptr, (float*)nullptr,
               (float*)nullptr);
  SESSD_CHECK_LAUNCH();
  const size_t quads = (size_t)batch * channels * (plane >> 2);
  SESSD_LAUNCH((bn2d_apply_kernel<false>), dim3((unsigned)((quads + NT - 1) / NT)), dim3(NT), 0, stream, x, (const float*)nullptr,
               (const float*)nullptr, channels, plane, gamma, beta, (const float*)save_mean, (const float*)save_invstd,
               (const float*)nullptr, (const float*)nullptr, 0.f, relu, y, quads, (const float*)nullptr);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// y == NULL: the ReLU mask from x (needs beta), as sessd_bn2d_relu_train_bwd_x
int sessd_bn2d_relu_train_bwd_stats(const float* dy, const float* x, const float* y, int batch, int channels, int plane,
                                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                    float* dgamma, float* dbeta, double* sums, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || channels > BN2D_MAX_CHANNELS || plane <= 0 || (plane & 3) || !dgamma || !dbeta || !sums)
    return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  unsigned* counters = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + BN2D_COUNTER_BYTES);
  BnFinal F{0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dgamma, dbeta, sums};
  SESSD_LAUNCH((bn2d_stats_kernel<1>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, gamma, beta,
               batch, channels, plane, relu, partial, counters, F);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_bn2d_relu_train_bwd_apply(const float* dy, const float* x, const float* y, int batch, int channels, int plane,
                                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                    const double* sums, const double* fwd_sums, float* dx, void* scratch, size_t scratch_bytes,
                                    hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || channels > BN2D_MAX_CHANNELS || plane <= 0 || (plane & 3) || !sums || !fwd_sums || !scratch)
    return SESSD_EINVAL;
  if (scratch_bytes < sessd_bn_sync_scratch_bytes(channels)) return SESSD_EWORKSPACE;
  float* gdb = (float*)scratch;
  float* gdg = gdb + channels;
  float* inv_n = gdg + channels;
  SESSD_LAUNCH((bn_sync_finalize_kernel<true>), dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, sums, fwd_sums, channels, 0.f, 0.f,
               (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, gdb, gdg, inv_n);
  SESSD_CHECK_LAUNCH();
  const size_t quads = (size_t)batch * channels * (plane >> 2);
  SESSD_LAUNCH((bn2d_apply_kernel<true>), dim3((unsigned)((quads + NT - 1) / NT)), dim3(NT), 0, stream, x, dy, y, channels, plane, gamma,
               beta, save_mean, save_invstd, (const float*)gdg, (const float*)gdb, 0.f, relu, dx, quads, (const float*)inv_n);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
