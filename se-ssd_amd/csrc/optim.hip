// Parameter update of the SE-SSD training step on gfx950, fused over FLAT float32
// buffers (3.8 M parameters = 15.2 MB): gradient-norm clip coefficient, decoupled weight decay, Adam, EMA teacher.
// Reference composition (host loops over ~100 parameter tensors, one tiny kernel each):
//   det3d/torchie/trainer/hooks/optimizer.py:50-53     clip_grad_norm_(max_norm=35, L2)
//   det3d/solver/fastai_optim.py:155-176               OptimWrapper.step(true_wd): p *= 1 - wd*lr ; opt.step() (Adam)
//   det3d/torchie/trainer/trainer_sessd.py:315-318     theta_T = alpha*theta_T + (1-alpha)*theta_S
// HBM-bound: one pass reads p, g, m, v, teacher and writes p, m, v, teacher = 36 B per parameter (137 MB per step),
// plus one 4 B/parameter read for the norm. The clip coefficient stays on the device (no host sync).
#include "common.hpp"

namespace {

constexpr int NORM_BLOCKS = 1024;

// deterministic two-stage sum of squares: fixed grid, grid-stride order per thread, wave + LDS tree per block
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  __shared__ double sm[4];
  double s = 0.0;
  const size_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)NORM_BLOCKS * 256) {
    const float4 x = g4[i];
    s += (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z + (double)x.w * x.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float x = g[(n4 << 2) + threadIdx.x];
    s += (double)x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// out[0] = ||g||_2, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0)
__global__ __launch_bounds__(64) void sumsq_final_kernel(const double* __restrict__ partial, float max_norm, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < NORM_BLOCKS; i += 64) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (threadIdx.x == 0) {
    const double norm = sqrt(s);
    out[0] = (float)norm;
    out[1] = max_norm > 0.f ? (float)fmin(1.0, (double)max_norm / (norm + 1e-6)) : 1.f;
  }
}

// plain sum of n floats, the same deterministic two-stage shape (double accumulation, fixed grid and order). Exists because
// torch's multi-block reductions clear a semaphore buffer with cudaMemsetAsync, and a hipMemsetAsync NODE of a replayed hipGraph
// writes garbage from the second replay on (ROCm 7.2 / gfx950; scripts/dbg_memset_graph.py, scripts/dbg_torch_graph_ops.py: a
// captured `x.mean()` of 1.4 M elements returns its first replay's value forever) -- a loss inside a captured training
// iteration must not reduce through torch.
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, size_t n, double* __restrict__ partial) {
  __shared__ double sm[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)NORM_BLOCKS * 256) s += (double)x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ __launch_bounds__(64) void sum_final_kernel(const double* __restrict__ partial, float scale, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < NORM_BLOCKS; i += 64) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (threadIdx.x == 0) out[0] = (float)(s * (double)scale);
}

struct AdamArgs {
  float decay;      // 1 - wd*lr
  float one_m_b1;   // 1 - beta1
  float beta2;
  float one_m_b2;
  float sqrt_bc2;   // sqrt(1 - beta2^t): a divisor, as in torch (denom = sqrt(v) / sqrt_bc2 + eps)
  float eps;
  float neg_step;   // -(lr / (1 - beta1^t))
  float alpha;      // EMA
  float one_m_alpha;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float* t, const AdamArgs& A, float coef) {
  g = coef < 1.f ? g * coef : g;
  p = p * A.decay;
  m = m + (g - m) * A.one_m_b1;
  v = v * A.beta2;
  v = v + A.one_m_b2 * g * g;
  const float denom = sqrtf(v) / A.sqrt_bc2 + A.eps;
  p = p + A.neg_step * (m / denom);
  if (t) *t = *t * A.alpha + A.one_m_alpha * p;
}

// Adev != nullptr: the step's constants come from device memory (written by one_cycle_args_kernel earlier on the stream), so
// that a captured graph replays with the CURRENT learning rate / momentum / bias corrections instead of the captured ones
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ teacher, size_t n, AdamArgs Ahost,
                                                        const AdamArgs* __restrict__ Adev, const float* __restrict__ clip2) {
  const AdamArgs A = Adev ? *Adev : Ahost;
  const float coef = clip2 ? clip2[1] : 1.f;
  const size_t n4 = n >> 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float4 tt = teacher ? reinterpret_cast<float4*>(teacher)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    adam_one(pp.x, gg.x, mm.x, vv.x, teacher ? &tt.x : nullptr, A, coef);
    adam_one(pp.y, gg.y, mm.y, vv.y, teacher ? &tt.y : nullptr, A, coef);
    adam_one(pp.z, gg.z, mm.z, vv.z, teacher ? &tt.z : nullptr, A, coef);
    adam_one(pp.w, gg.w, mm.w, vv.w, teacher ? &tt.w : nullptr, A, coef);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (teacher) reinterpret_cast<float4*>(teacher)[i] = tt;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t e = (n4 << 2) + threadIdx.x;
    adam_one(p[e], g[e], m[e], v[e], teacher ? teacher + e : nullptr, A, coef);
  }
}

__host__ __device__ inline AdamArgs make_adam_args(double lr, double weight_decay, double beta1, double beta2, double eps, int step,
                                                   double ema_alpha) {
  AdamArgs A;
  A.decay = (float)(1.0 - weight_decay * lr);
  A.one_m_b1 = (float)(1.0 - beta1);
  A.beta2 = (float)beta2;
  A.one_m_b2 = (float)(1.0 - beta2);
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  A.sqrt_bc2 = (float)sqrt(bc2);
  A.eps = (float)eps;
  A.neg_step = (float)(-(lr / bc1));
  A.alpha = (float)ema_alpha;
  A.one_m_alpha = (float)(1.0 - ema_alpha);
  return A;
}

// OneCycle (det3d/solver/learning_schedules_fastai.py:70-95, config.py:260) + the EMA coefficient (trainer_sessd.py:316) of
// iteration *global_step, on the device: lr cos-anneals low -> lr_max over the first pct_start of total_steps, then
// lr_max -> low / 1e4; the momentum mirrors it (mom_hi -> mom_lo -> mom_hi). Writes the Adam constants of that iteration
// (optimizer step t = *global_step + 1), lr_mom[0..1] = (lr, momentum), and advances *global_step.
__global__ void one_cycle_args_kernel(int* __restrict__ global_step, int total_steps, double lr_max, double mom_hi, double mom_lo,
                                      double div_factor, double pct_start, double weight_decay, double beta2, double eps,
                                      AdamArgs* __restrict__ out, float* __restrict__ lr_mom) {
  const int step = *global_step;
  const double pi = 3.141592653589793;
  const int a1 = (int)((double)total_steps * pct_start);
  const double low = lr_max / div_factor;
  double lr, mom;
  if (step < a1) {
    const double c = cos(pi * ((double)step / (double)a1)) + 1.0;
    lr = lr_max + (low - lr_max) / 2.0 * c;
    mom = mom_lo + (mom_hi - mom_lo) / 2.0 * c;
  } else {
    const double c = cos(pi * ((double)(step - a1) / (double)(total_steps - a1))) + 1.0;
    lr = low / 1e4 + (lr_max - low / 1e4) / 2.0 * c;
    mom = mom_hi + (mom_lo - mom_hi) / 2.0 * c;
  }
  const double alpha = fmin(1.0 - 1.0 / (double)(step + 1), 0.999);
  *out = make_adam_args(lr, weight_decay, mom, beta2, eps, step + 1, alpha);
  if (lr_mom) { lr_mom[0] = (float)lr; lr_mom[1] = (float)mom; }
  *global_step = step + 1;
}

}  // namespace

extern "C" {

size_t sessd_grad_clip_workspace_bytes(void) { return NORM_BLOCKS * sizeof(double); }

// out2 (device float[2]) = { ||grad||_2 , min(1, max_norm / (norm + 1e-6)) }; max_norm <= 0 disables clipping (coef 1).
int sessd_grad_clip_coef(const float* grad, size_t n, float max_norm, void* workspace, size_t workspace_bytes, float* out2,
                         hipStream_t stream) {
  if (workspace_bytes < sessd_grad_clip_workspace_bytes() || ((uintptr_t)grad & 15)) return SESSD_EINVAL;
  SESSD_LAUNCH(sumsq_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, stream, grad, n, (double*)workspace);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(sumsq_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, max_norm, out2);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// One fused pass over n float32 parameters (all pointers 16-byte aligned, device):
//   g' = grad * clip2[1] (if clip2 != NULL and < 1);  p *= 1 - wd*lr;  Adam(beta1, beta2, eps, step t >= 1) on (p, m, v);
//   ema_param = alpha*ema_param + (1-alpha)*p   (skipped when ema_param == NULL)
int sessd_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema_param, size_t n,
                        double lr, double weight_decay, double beta1, double beta2, double eps, int step,
                        const float* clip2, double ema_alpha, hipStream_t stream) {
  if (step < 1 || n == 0) return SESSD_EINVAL;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)ema_param) & 15)) return SESSD_EINVAL;
  const AdamArgs A = make_adam_args(lr, weight_decay, beta1, beta2, eps, step, ema_alpha);
  const size_t n4 = n >> 2;
  const unsigned blocks = (unsigned)((n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1);
  SESSD_LAUNCH(adam_ema_kernel, dim3(blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, ema_param, n, A,
               (const AdamArgs*)nullptr, clip2);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// out[0] = scale * sum of the n floats at x (deterministic; workspace of sessd_grad_clip_workspace_bytes()); no memset, no atomics
int sessd_sum_f32(const float* x, size_t n, float scale, void* workspace, size_t workspace_bytes, float* out, hipStream_t stream) {
  if (workspace_bytes < sessd_grad_clip_workspace_bytes() || !x || !out) return SESSD_EINVAL;
  SESSD_LAUNCH(sum_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, stream, x, n, (double*)workspace);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(sum_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, scale, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// The schedule on the device (so that the iteration can be a captured graph): reads and advances the device iteration counter
// *global_step, writes the nine Adam / EMA constants of that iteration to args9 (device float[9], consumed by
// sessd_adam_ema_step_dev) and (lr, momentum) to lr_mom2 (device float[2], may be NULL).
int sessd_one_cycle_args(int32_t* global_step, int total_steps, double lr_max, double mom_hi, double mom_lo, double div_factor,
                         double pct_start, double weight_decay, double beta2, double eps, float* args9, float* lr_mom2,
                         hipStream_t stream) {
  if (!global_step || !args9 || total_steps < 2 || !(pct_start > 0.0 && pct_start < 1.0) || div_factor <= 0.0) return SESSD_EINVAL;
  if ((int)((double)total_steps * pct_start) < 1) return SESSD_EINVAL;
  static_assert(sizeof(AdamArgs) == 9 * sizeof(float), "args9");
  SESSD_LAUNCH(one_cycle_args_kernel, dim3(1), dim3(1), 0, stream, global_step, total_steps, lr_max, mom_hi, mom_lo, div_factor,
               pct_start, weight_decay, beta2, eps, (AdamArgs*)args9, lr_mom2);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// sessd_adam_ema_step with its constants in device memory (args9 from sessd_one_cycle_args)
int sessd_adam_ema_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema_param, size_t n,
                            const float* args9, const float* clip2, hipStream_t stream) {
  if (n == 0 || !args9) return SESSD_EINVAL;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)ema_param) & 15)) return SESSD_EINVAL;
  const size_t n4 = n >> 2;
  const unsigned blocks = (unsigned)((n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1);
  AdamArgs unused = {};
  SESSD_LAUNCH(adam_ema_kernel, dim3(blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, ema_param, n, unused,
               (const AdamArgs*)args9, clip2);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
