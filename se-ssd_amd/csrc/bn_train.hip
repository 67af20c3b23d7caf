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
// STATUS: written in round 1 after the GPU budget was spent -- compiled, not yet run on hardware, not wired into the module
// path; tests/test_bn_train_gpu.py runs only with SESSD_EXPERIMENTAL=1 (vs torch.nn.BatchNorm1d + ReLU).
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
  dgamma[c] = (float)s1;
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
