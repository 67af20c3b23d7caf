// The attention tail of the SSFA neck in TRAIN mode, forward and backward (det3d/models/necks/rpn_v1.py:225-235 under the
// training step det3d/torchie/trainer/trainer_sessd.py:250-275):
//
//   s_k[p]  = sum_c w_k[c] * x_k[c][p]                 k = 0, 1     (w_0 / w_1: Conv2d(C, 1, 1, bias=False))
//   z_k[p]  = (s_k[p] - mean_k) * invstd_k * gamma_k + beta_k        (BatchNorm2d(1), batch statistics over all images and pixels)
//   a[p]    = softmax(z_0[p], z_1[p])
//   out[c][p] = x_0[c][p] * a_0[p] + x_1[c][p] * a_1[p]
//
// As torch modules this is 2 MIOpen convs (with layout transposes), 2 BatchNorms, a cat, a softmax, two multiplies and an add
// forward and their autograd chain backward: ~30 launches and ~20 passes over 72 MB maps per iteration (batch 4). Here:
//   forward   ssfa_dot_kernel<false>  : s_0, s_1 maps + their batch statistics (block partial sums in float64; the block that
//                                       finishes last adds them in block order and writes mean / invstd / running statistics)
//             ssfa_blend_kernel       : z, softmax, out                                   -- 2 reads of x_0, x_1, 1 write
//   backward  ssfa_dot_kernel<true>   : da_k[p] = sum_c g[c][p] x_k[c][p];  dz_0 = a_0 a_1 (da_0 - da_1) = -dz_1 (map) and
//                                       the three sums BatchNorm's backward needs (sum dz_0, sum dz_0 zhat_0, sum dz_0 zhat_1)
//                                       -> dgamma, dbeta
//             ssfa_bwd_apply_kernel   : ds_k = gamma_k invstd_k (dz_k - dbeta_k / N - zhat_k dgamma_k / N);
//                                       dx_k[c][p] = g[c][p] a_k[p] + w_k[c] ds_k[p];  dw_k[c] = sum_p ds_k[p] x_k[c][p]
//                                       (block = (channel, plane slice); the slice of a channel that finishes last adds the
//                                       channel's partials in slice order)                -- reads g, x_0, x_1 twice, writes dx_0, dx_1
// HBM-bound; deterministic (fixed summation orders, no float atomics). The arrival counters follow csrc/bn_train.hip: partial
// sums travel as agent-scope stores / loads, counters are zero on entry and zero on return.
#include "common.hpp"

namespace {

constexpr int NT = 256;
constexpr int SLICES = 16;                 // plane slices per channel in the backward apply
constexpr int MAX_CH = 1024;
constexpr size_t CTR_BYTES = 256 + (size_t)MAX_CH * 4;   // [0]: the dot kernels' counter; [64 ...]: one per channel

__device__ __forceinline__ void put_partial(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double get_partial(const double* p) {
  return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define SESSD_STORES_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

struct SsfaBn {           // the two BatchNorm2d(1) layers
  const float* gamma0; const float* beta0; const float* gamma1; const float* beta1;   // (1,) each; null: 1 / 0
  float* running_mean0; float* running_var0; float* running_mean1; float* running_var1;   // all or none
  float eps, momentum;
};

__device__ __forceinline__ float4 f4_fma(float4 a, float s, float4 c) {
  return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float4 f4_fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// softmax weight of branch 0 for one pixel from the two conv outputs (a_1 = 1 - a_0 is formed as e1 * inv, like torch.softmax)
struct Attn { float a0, a1, zh0, zh1; };
__device__ __forceinline__ Attn attention(float s0, float s1, const float* __restrict__ st, float g0, float b0, float g1, float b1) {
  Attn r;
  r.zh0 = (s0 - st[0]) * st[1];
  r.zh1 = (s1 - st[2]) * st[3];
  const float z0 = fmaf(r.zh0, g0, b0), z1 = fmaf(r.zh1, g1, b1);
  const float m = fmaxf(z0, z1);
  const float e0 = expf(z0 - m), e1 = expf(z1 - m);
  const float inv = 1.f / (e0 + e1);
  r.a0 = e0 * inv;
  r.a1 = e1 * inv;
  return r;
}

// thread = (4 consecutive pixels, channel quarter); block = 64 pixel quads. BWD false: dot with w_k; true: dot of g with x_k.
template <bool BWD>
__global__ __launch_bounds__(NT) void ssfa_dot_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                       const float* __restrict__ w0, const float* __restrict__ w1,
                                                       const float* __restrict__ g, int B, int C, int plane,
                                                       float* __restrict__ smap,          // fwd: out (2, B*plane); bwd: in
                                                       float* __restrict__ dzmap,         // bwd: out (B*plane)
                                                       float* stats,                      // fwd: out [mean0, invstd0, mean1, invstd1]
                                                       SsfaBn bn, float* dgamma, float* dbeta,   // bwd: out (2) each
                                                       double* partial, unsigned* counter) {
  __shared__ float4 part[2][4][64];
  __shared__ double red[4][NT];
  __shared__ int s_last;
  const int tid = threadIdx.x, pq = tid & 63, cq = tid >> 6;
  const int quads = plane >> 2;
  const long long total = (long long)B * quads;
  const long long Q = (long long)blockIdx.x * 64 + pq;
  const bool live = Q < total;
  const int b = live ? (int)(Q / quads) : 0, q = live ? (int)(Q - (long long)b * quads) : 0;
  const size_t base = (size_t)b * C * plane + 4 * (size_t)q;
  const int cper = C >> 2, c0 = cq * cper;
  float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
  if (live) {
#pragma unroll 4
    for (int c = c0; c < c0 + cper; ++c) {
      const size_t o = base + (size_t)c * plane;
      const float4 v0 = *reinterpret_cast<const float4*>(x0 + o), v1 = *reinterpret_cast<const float4*>(x1 + o);
      if (!BWD) {
        acc0 = f4_fma(v0, w0[c], acc0);
        acc1 = f4_fma(v1, w1[c], acc1);
      } else {
        const float4 gv = *reinterpret_cast<const float4*>(g + o);
        acc0 = f4_fma4(v0, gv, acc0);
        acc1 = f4_fma4(v1, gv, acc1);
      }
    }
  }
  part[0][cq][pq] = acc0;
  part[1][cq][pq] = acc1;
  __syncthreads();
  double t[4] = {0.0, 0.0, 0.0, 0.0};
  if (cq == 0 && live) {   // quarter order 0..3
    float r0[4], r1[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 p0 = part[k][0][pq], p1 = part[k][1][pq], p2 = part[k][2][pq], p3 = part[k][3][pq];
      float* r = k ? r1 : r0;
      r[0] = ((p0.x + p1.x) + p2.x) + p3.x; r[1] = ((p0.y + p1.y) + p2.y) + p3.y;
      r[2] = ((p0.z + p1.z) + p2.z) + p3.z; r[3] = ((p0.w + p1.w) + p2.w) + p3.w;
    }
    const size_t po = (size_t)Q * 4;
    const size_t n_all = (size_t)total * 4;
    if (!BWD) {
      *reinterpret_cast<float4*>(smap + po) = make_float4(r0[0], r0[1], r0[2], r0[3]);
      *reinterpret_cast<float4*>(smap + n_all + po) = make_float4(r1[0], r1[1], r1[2], r1[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[0] += (double)r0[e]; t[1] += (double)r0[e] * (double)r0[e];
        t[2] += (double)r1[e]; t[3] += (double)r1[e] * (double)r1[e];
      }
    } else {
      const float4 s0 = *reinterpret_cast<const float4*>(smap + po), s1 = *reinterpret_cast<const float4*>(smap + n_all + po);
      const float sa[4] = {s0.x, s0.y, s0.z, s0.w}, sb[4] = {s1.x, s1.y, s1.z, s1.w};
      const float g0 = bn.gamma0 ? bn.gamma0[0] : 1.f, b0 = bn.beta0 ? bn.beta0[0] : 0.f;
      const float g1 = bn.gamma1 ? bn.gamma1[0] : 1.f, b1 = bn.beta1 ? bn.beta1[0] : 0.f;
      float dz[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Attn A = attention(sa[e], sb[e], stats, g0, b0, g1, b1);
        dz[e] = A.a0 * A.a1 * (r0[e] - r1[e]);
        t[0] += (double)dz[e];
        t[1] += (double)dz[e] * (double)A.zh0;
        t[2] += (double)dz[e] * (double)A.zh1;
      }
      *reinterpret_cast<float4*>(dzmap + po) = make_float4(dz[0], dz[1], dz[2], dz[3]);
    }
  }
  // block sums (wave 0 holds them): lane tree in LDS, fixed order
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][tid] = t[k];
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
    }
    __syncthreads();
  }
  if (tid < 4) put_partial(partial + (size_t)blockIdx.x * 4 + tid, red[tid][0]);
  SESSD_STORES_DONE();
  __syncthreads();
  if (tid == 0) s_last = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  // ---- the last block: all block partials in block order (thread t takes blocks t, t + 256, ...; then a fixed tree)
  double u[4] = {0.0, 0.0, 0.0, 0.0};
  for (int blk = tid; blk < (int)gridDim.x; blk += NT) {
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] += get_partial(partial + (size_t)blk * 4 + k);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][tid] = u[k];
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
    }
    __syncthreads();
  }
  if (tid != 0) return;
  __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double n = (double)total * 4.0;
  if (!BWD) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double m = red[2 * k][0] / n;
      double var = red[2 * k + 1][0] / n - m * m;
      if (var < 0.0) var = 0.0;
      stats[2 * k] = (float)m;
      stats[2 * k + 1] = (float)(1.0 / sqrt(var + (double)bn.eps));
      float* rm = k ? bn.running_mean1 : bn.running_mean0;
      float* rv = k ? bn.running_var1 : bn.running_var0;
      if (rm) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        rm[0] = (float)((1.0 - bn.momentum) * rm[0] + bn.momentum * m);
        rv[0] = (float)((1.0 - bn.momentum) * rv[0] + bn.momentum * unbiased);
      }
    }
  } else {   // dz_1 = -dz_0
    dbeta[0] = (float)red[0][0];
    dbeta[1] = (float)(-red[0][0]);
    dgamma[0] = (float)red[1][0];
    dgamma[1] = (float)(-red[2][0]);
  }
}

// thread = 4 consecutive pixels, one slice of the channels (blockIdx.y)
__global__ __launch_bounds__(NT) void ssfa_blend_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                         const float* __restrict__ smap, const float* __restrict__ stats,
                                                         SsfaBn bn, int B, int C, int plane, int cper, float* __restrict__ out) {
  const int quads = plane >> 2;
  const long long total = (long long)B * quads;
  const long long Q = (long long)blockIdx.x * NT + threadIdx.x;
  if (Q >= total) return;
  const int b = (int)(Q / quads), q = (int)(Q - (long long)b * quads);
  const size_t po = (size_t)Q * 4, n_all = (size_t)total * 4;
  const float4 s0 = *reinterpret_cast<const float4*>(smap + po), s1 = *reinterpret_cast<const float4*>(smap + n_all + po);
  const float g0 = bn.gamma0 ? bn.gamma0[0] : 1.f, b0 = bn.beta0 ? bn.beta0[0] : 0.f;
  const float g1 = bn.gamma1 ? bn.gamma1[0] : 1.f, b1 = bn.beta1 ? bn.beta1[0] : 0.f;
  const Attn A0 = attention(s0.x, s1.x, stats, g0, b0, g1, b1), A1 = attention(s0.y, s1.y, stats, g0, b0, g1, b1);
  const Attn A2 = attention(s0.z, s1.z, stats, g0, b0, g1, b1), A3 = attention(s0.w, s1.w, stats, g0, b0, g1, b1);
  const int c0 = blockIdx.y * cper, c1 = min(C, c0 + cper);
  const size_t base = (size_t)b * C * plane + 4 * (size_t)q;
#pragma unroll 4
  for (int c = c0; c < c1; ++c) {
    const size_t o = base + (size_t)c * plane;
    const float4 v0 = *reinterpret_cast<const float4*>(x0 + o), v1 = *reinterpret_cast<const float4*>(x1 + o);
    float4 r;
    r.x = v0.x * A0.a0 + v1.x * A0.a1; r.y = v0.y * A1.a0 + v1.y * A1.a1;
    r.z = v0.z * A2.a0 + v1.z * A2.a1; r.w = v0.w * A3.a0 + v1.w * A3.a1;
    *reinterpret_cast<float4*>(out + o) = r;
  }
}

// block (c, slice): dx_0, dx_1 of channel c over the slice's pixels of every image, and the slice's part of dw_0[c], dw_1[c]
__global__ __launch_bounds__(NT) void ssfa_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x0,
                                                             const float* __restrict__ x1, const float* __restrict__ w0,
                                                             const float* __restrict__ w1, const float* __restrict__ smap,
                                                             const float* __restrict__ dzmap, const float* __restrict__ stats,
                                                             SsfaBn bn, const float* __restrict__ dgamma,
                                                             const float* __restrict__ dbeta, int B, int C, int plane,
                                                             float* __restrict__ dx0, float* __restrict__ dx1, float* dw0, float* dw1,
                                                             double* partial, unsigned* counters) {
  __shared__ double sm[2][NT / 64];
  const int c = blockIdx.x, s = blockIdx.y;
  const int quads = plane >> 2;
  const int chunk = sessd_divup(quads, SLICES);
  const int q0 = s * chunk, q1 = min(quads, q0 + chunk);
  const size_t n_all = (size_t)B * plane;
  const float inv_n = 1.f / (float)n_all;
  const float g0 = bn.gamma0 ? bn.gamma0[0] : 1.f, b0 = bn.beta0 ? bn.beta0[0] : 0.f;
  const float g1 = bn.gamma1 ? bn.gamma1[0] : 1.f, b1 = bn.beta1 ? bn.beta1[0] : 0.f;
  const float k0 = g0 * stats[1], k1 = g1 * stats[3];
  const float db0 = dbeta[0] * inv_n, db1 = dbeta[1] * inv_n, dg0 = dgamma[0] * inv_n, dg1 = dgamma[1] * inv_n;
  const float wc0 = w0[c], wc1 = w1[c];
  double t0 = 0.0, t1 = 0.0;
  for (int b = 0; b < B; ++b) {
    const size_t base = ((size_t)b * C + c) * plane, pbase = (size_t)b * plane;
    for (int q = q0 + (int)threadIdx.x; q < q1; q += NT) {
      const size_t o = base + 4 * (size_t)q, po = pbase + 4 * (size_t)q;
      const float4 gv = *reinterpret_cast<const float4*>(g + o);
      const float4 v0 = *reinterpret_cast<const float4*>(x0 + o), v1 = *reinterpret_cast<const float4*>(x1 + o);
      const float4 s0 = *reinterpret_cast<const float4*>(smap + po), s1 = *reinterpret_cast<const float4*>(smap + n_all + po);
      const float4 dz = *reinterpret_cast<const float4*>(dzmap + po);
      const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, xa[4] = {v0.x, v0.y, v0.z, v0.w}, xb[4] = {v1.x, v1.y, v1.z, v1.w};
      const float sa[4] = {s0.x, s0.y, s0.z, s0.w}, sb[4] = {s1.x, s1.y, s1.z, s1.w}, dza[4] = {dz.x, dz.y, dz.z, dz.w};
      float ra[4], rb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Attn A = attention(sa[e], sb[e], stats, g0, b0, g1, b1);
        const float ds0 = k0 * (dza[e] - db0 - A.zh0 * dg0);
        const float ds1 = k1 * (-dza[e] - db1 - A.zh1 * dg1);
        ra[e] = fmaf(wc0, ds0, ga[e] * A.a0);
        rb[e] = fmaf(wc1, ds1, ga[e] * A.a1);
        t0 += (double)ds0 * (double)xa[e];
        t1 += (double)ds1 * (double)xb[e];
      }
      *reinterpret_cast<float4*>(dx0 + o) = make_float4(ra[0], ra[1], ra[2], ra[3]);
      *reinterpret_cast<float4*>(dx1 + o) = make_float4(rb[0], rb[1], rb[2], rb[3]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    t0 += __shfl_xor(t0, o, 64);
    t1 += __shfl_xor(t1, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = t0; sm[1][threadIdx.x >> 6] = t1; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  t0 = 0.0; t1 = 0.0;
  for (int w = 0; w < NT / 64; ++w) { t0 += sm[0][w]; t1 += sm[1][w]; }
  double* pc = partial + (size_t)c * SLICES * 2;
  put_partial(pc + s * 2 + 0, t0);
  put_partial(pc + s * 2 + 1, t1);
  SESSD_STORES_DONE();
  if (__hip_atomic_fetch_add(counters + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != SLICES - 1) return;
  double a0[SLICES], a1[SLICES];
#pragma unroll
  for (int k = 0; k < SLICES; ++k) { a0[k] = get_partial(pc + k * 2 + 0); a1[k] = get_partial(pc + k * 2 + 1); }
  t0 = 0.0; t1 = 0.0;
#pragma unroll
  for (int k = 0; k < SLICES; ++k) { t0 += a0[k]; t1 += a1[k]; }
  __hip_atomic_store(counters + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  dw0[c] = (float)t0;
  dw1[c] = (float)t1;
}

bool shape_ok(int batch, int channels, int plane) {
  return batch >= 1 && channels >= 4 && channels <= MAX_CH && (channels & 3) == 0 && plane >= 4 && (plane & 3) == 0 &&
         (size_t)batch * channels * plane * 4 < 0x7FFFFFFF00ull;
}
int dot_blocks(int batch, int plane) { return (int)(((long long)batch * (plane >> 2) + 63) / 64); }

}  // namespace

extern "C" {

// leading CTR_BYTES: arrival counters (zero on entry, zero on return), then the block partials of the larger of the two passes
size_t sessd_ssfa_fuse_train_workspace_bytes(int batch, int channels, int plane) {
  if (!shape_ok(batch, channels, plane)) return 0;
  const size_t dot = (size_t)dot_blocks(batch, plane) * 4 * sizeof(double);
  const size_t app = (size_t)channels * SLICES * 2 * sizeof(double);
  return CTR_BYTES + (dot > app ? dot : app);
}

// Forward. x0, x1, out (batch, channels, plane = H * W) float32, channels % 4 == 0, plane % 4 == 0; w0, w1 (channels);
// gamma / beta / running_* of the two BatchNorm2d(1): one float each (gamma, beta may be NULL = 1, 0; running_*: all four or
// none, updated in place with `momentum`, unbiased variance). Saved for the backward: smap (2, batch * plane) = the two conv
// outputs, stats (4) = [mean0, invstd0, mean1, invstd1].
int sessd_ssfa_fuse_train_fwd(const float* x0, const float* x1, int batch, int channels, int plane, const float* w0, const float* w1,
                              const float* gamma0, const float* beta0, const float* gamma1, const float* beta1, float eps,
                              float momentum, float* running_mean0, float* running_var0, float* running_mean1, float* running_var1,
                              float* out, float* smap, float* stats, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!shape_ok(batch, channels, plane) || !x0 || !x1 || !w0 || !w1 || !out || !smap || !stats) return SESSD_EINVAL;
  const int nrun = (running_mean0 != nullptr) + (running_var0 != nullptr) + (running_mean1 != nullptr) + (running_var1 != nullptr);
  if (nrun != 0 && nrun != 4) return SESSD_EINVAL;
  if (workspace_bytes < sessd_ssfa_fuse_train_workspace_bytes(batch, channels, plane)) return SESSD_EWORKSPACE;
  unsigned* counters = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + CTR_BYTES);
  SsfaBn bn{gamma0, beta0, gamma1, beta1, running_mean0, running_var0, running_mean1, running_var1, eps, momentum};
  SESSD_LAUNCH((ssfa_dot_kernel<false>), dim3(dot_blocks(batch, plane)), dim3(NT), 0, stream, x0, x1, w0, w1, (const float*)nullptr,
               batch, channels, plane, smap, (float*)nullptr, stats, bn, (float*)nullptr, (float*)nullptr, partial, counters);
  SESSD_CHECK_LAUNCH();
  const long long quads = (long long)batch * (plane >> 2);
  const int cper = 16;
  SESSD_LAUNCH(ssfa_blend_kernel, dim3((unsigned)((quads + NT - 1) / NT), sessd_divup(channels, cper)), dim3(NT), 0, stream, x0, x1,
               (const float*)smap, (const float*)stats, bn, batch, channels, plane, cper, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// Backward: grad_out (batch, channels, plane) -> dx0, dx1 (same shape), dw0, dw1 (channels), dgamma (2), dbeta (2) = the
// gradients of [gamma0, gamma1] / [beta0, beta1]. smap / stats from the forward; dzmap (batch * plane) is scratch.
int sessd_ssfa_fuse_train_bwd(const float* grad_out, const float* x0, const float* x1, int batch, int channels, int plane,
                              const float* w0, const float* w1, const float* gamma0, const float* beta0, const float* gamma1,
                              const float* beta1, const float* smap, const float* stats, float* dzmap, float* dx0, float* dx1,
                              float* dw0, float* dw1, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                              hipStream_t stream) {
  if (!shape_ok(batch, channels, plane) || !grad_out || !x0 || !x1 || !w0 || !w1 || !smap || !stats || !dzmap || !dx0 || !dx1 ||
      !dw0 || !dw1 || !dgamma || !dbeta)
    return SESSD_EINVAL;
  if (workspace_bytes < sessd_ssfa_fuse_train_workspace_bytes(batch, channels, plane)) return SESSD_EWORKSPACE;
  unsigned* counters = (unsigned*)workspace;
  double* partial = (double*)((char*)workspace + CTR_BYTES);
  SsfaBn bn{gamma0, beta0, gamma1, beta1, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
  SESSD_LAUNCH((ssfa_dot_kernel<true>), dim3(dot_blocks(batch, plane)), dim3(NT), 0, stream, x0, x1, w0, w1, grad_out, batch, channels,
               plane, const_cast<float*>(smap), dzmap, const_cast<float*>(stats), bn, dgamma, dbeta, partial, counters);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(ssfa_bwd_apply_kernel, dim3(channels, SLICES), dim3(NT), 0, stream, grad_out, x0, x1, w0, w1, smap, (const float*)dzmap,
               stats, bn, (const float*)dgamma, (const float*)dbeta, batch, channels, plane, dx0, dx1, dw0, dw1, partial, counters + 64);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
