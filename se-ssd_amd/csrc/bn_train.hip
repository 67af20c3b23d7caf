// Train-mode BatchNorm1d + ReLU over a sparse level's feature table (N rows x C channels), forward and backward
// (DESIGN.md section 9 item 2). In the SE-SSD training step (det3d/torchie/trainer/trainer_sessd.py:250-275) both networks run
// SpMiddleFHD in train mode (scn.py:103-148: BatchNorm1d(eps=1e-3, momentum=0.01) + ReLU after each of the 14 sparse convs), which
// as torch modules costs ~5 launches forward and ~6 backward per layer on tables of 3 k - 60 k rows: launch-bound. Here:
//   forward   stats (partial sums per row chunk, double)  ->  finalise (mean, 1/sqrt(var + eps), running statistics)
//             ->  y = max(0, (x - mean) * invstd * gamma + beta)
//   backward  dz = dy * [y > 0];  partial sums of dz and dz * xhat  ->  dgamma, dbeta  ->
//             dx = gamma * invstd * (dz - dbeta / N - xhat * dgamma / N)
// The row count N stays on the device. Reductions are deterministic: a fixed grid of row chunks, partials summed in order.
// HBM-bound elementwise work (3 passes over x forward, 3 backward).
//
//
// Second half of the file: the same for the dense BEV layout (B, C, H, W) of the SSFA neck in train mode
// (det3d/models/necks/rpn_v1.py:131-210: BatchNorm2d(eps=1e-3, momentum=0.01) + ReLU after each of its 13 convolutions) --
// one block per (channel, plane slice), 16-byte accesses along the plane, the same three passes.
// tests/test_bn_train_gpu.py compares both with torch.nn.BatchNorm1d / BatchNorm2d + ReLU (forward, running statistics, gradients).
#include "common.hpp"

namespace {

constexpr int BN_BLOCKS = 128;
constexpr int NT = 256;

// two per-channel sums over the rows of this block's chunk; thread = (row lane, channel); C in {4, 8, 16, 32, 64, 128}
template <bool BWD>
__global__ __launch_bounds__(NT) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const int* __restrict__ n_dev,
                                                         int n_cap, int C, int relu, double* __restrict__ partial) {
  __shared__ double sm[2][NT];
  const int n = min(n_dev[0], n_cap);
  const int lanes = NT / C;                 // row lanes per block
  const int c = threadIdx.x % C, rl = threadIdx.x / C;
  const int chunk = sessd_divup(n, BN_BLOCKS);
  const int r0 = blockIdx.x * chunk, r1 = min(n, r0 + chunk);
  double s0 = 0.0, s1 = 0.0;
  float mu = 0.f, is = 0.f;
  if (BWD) { mu = mean[c]; is = invstd[c]; }
  for (int r = r0 + rl; r < r1; r += lanes) {
    const size_t o = (size_t)r * C + c;
    if (!BWD) {
      const double v = x[o];
      s0 += v;
      s1 += v * v;
    } else {
      float dz = dy[o];
      if (relu && !(y[o] > 0.f)) dz = 0.f;
      s0 += (double)dz;
      s1 += (double)dz * (double)((x[o] - mu) * is);
    }
  }
  sm[0][threadIdx.x] = s0;
  sm[1][threadIdx.x] = s1;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < lanes; ++l) {
      s0 += sm[0][l * C + c];
      s1 += sm[1][l * C + c];
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c] = s0;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = s1;
  }
}

__global__ void bn_fwd_final_kernel(const double* __restrict__ partial, const int* __restrict__ n_dev, int n_cap, int C, float eps,
                                    float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                    float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int n = min(n_dev[0], n_cap);
  double s0 = 0.0, s1 = 0.0;
  for (int b = 0; b < BN_BLOCKS; ++b) {
    s0 += partial[((size_t)b * 2 + 0) * C + c];
    s1 += partial[((size_t)b * 2 + 1) * C + c];
  }
  const double mean = n > 0 ? s0 / n : 0.0;
  double var = n > 0 ? s1 / n - mean * mean : 0.0;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean && n > 0) {   // torch: running = (1 - momentum) * running + momentum * batch, variance unbiased
    const double unbiased = n > 1 ? var * n / (n - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

__global__ __launch_bounds__(NT) void bn_fwd_apply_kernel(const float* __restrict__ x, const int* __restrict__ n_dev, int n_cap,
                                                           int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           int relu, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  const size_t total = (size_t)min(n_dev[0], n_cap) * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  float v = (x[i] - mean[c]) * invstd[c] * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
  if (relu) v = fmaxf(v, 0.f);
  y[i] = v;
}

__global__ void bn_bwd_final_kernel(const double* __restrict__ partial, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int b = 0; b < BN_BLOCKS; ++b) {
    s0 += partial[((size_t)b * 2 + 0) * C + c];
    s1 += partial[((size_t)b * 2 + 1) * C + c];
  }
  dbeta[c] = (float)s0;
  if (dgamma) dgamma[c] = (float)s1;
}

__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const int* __restrict__ n_dev, int n_cap,
                                                           int C, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, int relu, float* __restrict__ dx) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  const int n = min(n_dev[0], n_cap);
  if (i >= (size_t)n * C) return;
  const int c = (int)(i % C);
  float dz = dy[i];
  if (relu && !(y[i] > 0.f)) dz = 0.f;
  const float xhat = (x[i] - mean[c]) * invstd[c];
  const float inv_n = 1.f / (float)n;
  dx[i] = (gamma ? gamma[c] : 1.f) * invstd[c] * (dz - dbeta[c] * inv_n - xhat * dgamma[c] * inv_n);
}

bool channels_ok(int C) { return C >= 1 && C <= NT && NT % C == 0; }

}  // namespace

extern "C" {

size_t sessd_bn_relu_train_workspace_bytes(int channels) { return (size_t)BN_BLOCKS * 2 * channels * sizeof(double); }

// y = relu?( (x - mean) * invstd * gamma + beta ) with the batch statistics of rows < *n_dev of x (n_cap, channels);
// save_mean / save_invstd (channels) for the backward; running_mean / running_var updated in place when not NULL
// (momentum, unbiased variance: torch.nn.BatchNorm1d semantics). channels must divide 256.
int sessd_bn_relu_train_fwd(const float* x, const int* n_dev, int n_cap, int channels, const float* gamma, const float* beta,
                            float eps, float momentum, int relu, float* running_mean, float* running_var, float* y,
                            float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels)) return SESSD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  double* partial = (double*)workspace;
  SESSD_LAUNCH((bn_partial_kernel<false>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, (const float*)nullptr, (const float*)nullptr,
               (const float*)nullptr, (const float*)nullptr, n_dev, n_cap, channels, 0, partial);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(bn_fwd_final_kernel, dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, partial, n_dev, n_cap, channels, eps,
               momentum, running_mean, running_var, save_mean, save_invstd);
  SESSD_CHECK_LAUNCH();
  const size_t total = (size_t)n_cap * channels;
  SESSD_LAUNCH(bn_fwd_apply_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, stream, x, n_dev, n_cap, channels, gamma,
               beta, save_mean, save_invstd, relu, y);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// gradients of the above: dx (n_cap, channels; rows < *n_dev written), dgamma, dbeta (channels). y is the forward output
// (the ReLU mask), x the forward input.
int sessd_bn_relu_train_bwd(const float* dy, const float* x, const float* y, const int* n_dev, int n_cap, int channels,
                            const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dx,
                            float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n_cap <= 0 || !channels_ok(channels)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  double* partial = (double*)workspace;
  SESSD_LAUNCH((bn_partial_kernel<true>), dim3(BN_BLOCKS), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, n_dev, n_cap,
               channels, relu, partial);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(bn_bwd_final_kernel, dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, partial, channels, dgamma, dbeta);
  SESSD_CHECK_LAUNCH();
  const size_t total = (size_t)n_cap * channels;
  SESSD_LAUNCH(bn_bwd_apply_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, stream, dy, x, y, n_dev, n_cap, channels,
               gamma, save_mean, save_invstd, dgamma, dbeta, relu, dx);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ dense (B, C, H, W) layout
namespace {

constexpr int BN2D_SPLIT = 16;   // plane slices per channel: C x 16 blocks of partial sums

// block (c, s): the two sums over the pixels [s * chunk, (s + 1) * chunk) of channel c in every image; plane % 4 == 0
template <bool BWD>
__global__ __launch_bounds__(NT) void bn2d_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, int B, int C, int plane, int relu,
                                                           double* __restrict__ partial) {
  __shared__ double sm[2][NT / 64];
  const int c = blockIdx.x, s = blockIdx.y;
  const int quads = plane >> 2;
  const int chunk = sessd_divup(quads, BN2D_SPLIT);
  const int q0 = s * chunk, q1 = min(quads, q0 + chunk);
  double s0 = 0.0, s1 = 0.0;
  float mu = 0.f, is = 0.f;
  if (BWD) { mu = mean[c]; is = invstd[c]; }
  for (int b = 0; b < B; ++b) {
    const size_t base = ((size_t)b * C + c) * plane;
    for (int q = q0 + (int)threadIdx.x; q < q1; q += NT) {
      const float4 xv = *reinterpret_cast<const float4*>(x + base + 4 * (size_t)q);
      if (!BWD) {
        s0 += (double)xv.x + (double)xv.y + (double)xv.z + (double)xv.w;
        s1 += (double)xv.x * xv.x + (double)xv.y * xv.y + (double)xv.z * xv.z + (double)xv.w * xv.w;
      } else {
        float4 dz = *reinterpret_cast<const float4*>(dy + base + 4 * (size_t)q);
        if (relu) {
          const float4 yv = *reinterpret_cast<const float4*>(y + base + 4 * (size_t)q);
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
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int w = 0; w < NT / 64; ++w) { t0 += sm[0][w]; t1 += sm[1][w]; }
    partial[((size_t)c * BN2D_SPLIT + s) * 2 + 0] = t0;
    partial[((size_t)c * BN2D_SPLIT + s) * 2 + 1] = t1;
  }
}

__global__ void bn2d_fwd_final_kernel(const double* __restrict__ partial, long long n, int C, float eps, float momentum,
                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                      float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int s = 0; s < BN2D_SPLIT; ++s) {
    s0 += partial[((size_t)c * BN2D_SPLIT + s) * 2 + 0];
    s1 += partial[((size_t)c * BN2D_SPLIT + s) * 2 + 1];
  }
  const double mean = s0 / (double)n;
  double var = s1 / (double)n - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

__global__ void bn2d_bwd_final_kernel(const double* __restrict__ partial, int C, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int s = 0; s < BN2D_SPLIT; ++s) {
    s0 += partial[((size_t)c * BN2D_SPLIT + s) * 2 + 0];
    s1 += partial[((size_t)c * BN2D_SPLIT + s) * 2 + 1];
  }
  dbeta[c] = (float)s0;
  if (dgamma) dgamma[c] = (float)s1;  // null: only the plain channel sums are wanted (sessd_nchw_channel_sum)
}

// one thread = four consecutive pixels of one (image, channel) plane
template <bool BWD>
__global__ __launch_bounds__(NT) void bn2d_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ y_in, int C, int plane,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                         float inv_n, int relu, float* __restrict__ out, size_t total_quads) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= total_quads) return;
  const int c = (int)((i / (size_t)(plane >> 2)) % (size_t)C);
  const float mu = mean[c], is = invstd[c], g = gamma ? gamma[c] : 1.f;
  const float4 xv = *reinterpret_cast<const float4*>(x + 4 * i);
  float4 r;
  if (!BWD) {
    const float bt = beta ? beta[c] : 0.f;
    r.x = (xv.x - mu) * is * g + bt; r.y = (xv.y - mu) * is * g + bt; r.z = (xv.z - mu) * is * g + bt; r.w = (xv.w - mu) * is * g + bt;
    if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
  } else {
    float4 dz = *reinterpret_cast<const float4*>(dy + 4 * i);
    if (relu) {
      const float4 yv = *reinterpret_cast<const float4*>(y_in + 4 * i);
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

size_t sessd_bn2d_relu_train_workspace_bytes(int channels) { return (size_t)channels * BN2D_SPLIT * 2 * sizeof(double); }

// out[c] = sum over images and pixels of x[b][c][.] (the bias gradient of a conv: det3d's heads): the statistics pass of the
// train-mode BatchNorm above with only its first sum kept -- deterministic, no atomics, no memset (a torch reduction of this
// size clears a semaphore buffer with a memset, which a replayed hipGraph does not execute correctly on this stack).
int sessd_nchw_channel_sum(const float* x, int batch, int channels, int plane, float* out, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
  if (batch < 1 || channels < 1 || plane < 4 || (plane & 3)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  double* partial = (double*)workspace;
  SESSD_LAUNCH((bn2d_partial_kernel<false>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, (const float*)nullptr,
               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, batch, channels, plane, 0, partial);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(bn2d_bwd_final_kernel, dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, partial, channels, (float*)nullptr, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// BatchNorm2d (train mode) + optional ReLU on x (batch, channels, plane = H * W; plane % 4 == 0), torch.nn.BatchNorm2d semantics
// (biased batch variance to normalise, unbiased into running_var); save_mean / save_invstd (channels) for the backward.
int sessd_bn2d_relu_train_fwd(const float* x, int batch, int channels, int plane, const float* gamma, const float* beta, float eps,
                              float momentum, int relu, float* running_mean, float* running_var, float* y, float* save_mean,
                              float* save_invstd, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || plane <= 0 || (plane & 3)) return SESSD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  double* partial = (double*)workspace;
  SESSD_LAUNCH((bn2d_partial_kernel<false>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, (const float*)nullptr,
               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, batch, channels, plane, 0, partial);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(bn2d_fwd_final_kernel, dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, partial, (long long)batch * plane,
               channels, eps, momentum, running_mean, running_var, save_mean, save_invstd);
  SESSD_CHECK_LAUNCH();
  const size_t quads = (size_t)batch * channels * (plane >> 2);
  SESSD_LAUNCH((bn2d_apply_kernel<false>), dim3((unsigned)((quads + NT - 1) / NT)), dim3(NT), 0, stream, x, (const float*)nullptr,
               (const float*)nullptr, channels, plane, gamma, beta, save_mean, save_invstd, (const float*)nullptr,
               (const float*)nullptr, 0.f, relu, y, quads);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// gradients of the above: dx (same shape), dgamma, dbeta (channels); y = the forward output (ReLU mask), x = the forward input
int sessd_bn2d_relu_train_bwd(const float* dy, const float* x, const float* y, int batch, int channels, int plane,
                              const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dx,
                              float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (batch <= 0 || channels <= 0 || plane <= 0 || (plane & 3)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_bn2d_relu_train_workspace_bytes(channels)) return SESSD_EWORKSPACE;
  double* partial = (double*)workspace;
  SESSD_LAUNCH((bn2d_partial_kernel<true>), dim3(channels, BN2D_SPLIT), dim3(NT), 0, stream, x, dy, y, save_mean, save_invstd, batch,
               channels, plane, relu, partial);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(bn2d_bwd_final_kernel, dim3(sessd_divup(channels, 64)), dim3(64), 0, stream, partial, channels, dgamma, dbeta);
  SESSD_CHECK_LAUNCH();
  const size_t quads = (size_t)batch * channels * (plane >> 2);
  SESSD_LAUNCH((bn2d_apply_kernel<true>), dim3((unsigned)((quads + NT - 1) / NT)), dim3(NT), 0, stream, x, dy, y, channels, plane,
               gamma, (const float*)nullptr, save_mean, save_invstd, dgamma, dbeta, 1.f / (float)((long long)batch * plane), relu, dx,
               quads);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
