// SparseConvTensor.dense() of spconv v1 (det3d/models/backbones/scn.py:184: `ret = ret.dense()`) and its gradient:
//   dense[b][c][z][y][x] = features[i][c]   for site i = (b,z,y,x);   grad_features[i][c] = grad_dense[b][c][z][y][x]
// (the inference engine never materialises the site table of the last level: its last sparse conv scatters straight into the
// BEV map, sparse_conv.hip DENSE_OUT; this is the module-path / training twin). One thread per (site, channel), channel
// fastest: feature reads / gradient writes are coalesced, the dense side is touched once per element.
#include "common.hpp"

namespace {

// n_dev != nullptr: the table has capacity `n` rows of which the first *n_dev are sites (rows beyond: untouched by the scatter,
// zero in the gathered gradient)
template <bool GATHER>
__global__ __launch_bounds__(256) void sparse_dense_kernel(float* __restrict__ feat, const int* __restrict__ indices, int n,
                                                            const int* __restrict__ n_dev, int channels, int D, int H, int W,
                                                            float* __restrict__ dense) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (size_t)n * channels) return;
  if (n_dev && id >= (size_t)min(n_dev[0], n) * channels) {
    if (GATHER) feat[id] = 0.f;
    return;
  }
  const int i = (int)(id / channels), c = (int)(id - (size_t)i * channels);
  const int4 s = *reinterpret_cast<const int4*>(indices + (size_t)i * 4);
  const size_t o = ((((size_t)s.x * channels + c) * D + s.y) * H + s.z) * W + s.w;
  if (GATHER)
    feat[id] = dense[o];
  else
    dense[o] = feat[id];
}

}  // namespace

extern "C" {

// features (n, channels) -> dense (batch, channels, D, H, W), pre-zeroed by the caller; indices (n,4) [b,z,y,x] unique
int sessd_sparse_to_dense_dev(const float* features, const int* indices, int n_cap, const int* n_dev, int channels,
                              const int* dims3, float* dense, hipStream_t stream) {
  if (n_cap < 0 || channels <= 0 || !dims3) return SESSD_EINVAL;
  if (n_cap == 0) return SESSD_OK;
  const size_t total = (size_t)n_cap * channels;
  SESSD_LAUNCH((sparse_dense_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
               const_cast<float*>(features), indices, n_cap, n_dev, channels, dims3[0], dims3[1], dims3[2], dense);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_sparse_to_dense(const float* features, const int* indices, int n, int channels, const int* dims3, float* dense,
                          hipStream_t stream) {
  return sessd_sparse_to_dense_dev(features, indices, n, nullptr, channels, dims3, dense, stream);
}

// the gradient of the above: grad_features (n, channels) <- grad_dense at the sites
int sessd_dense_to_sparse_dev(const float* dense, const int* indices, int n_cap, const int* n_dev, int channels,
                              const int* dims3, float* features, hipStream_t stream) {
  if (n_cap < 0 || channels <= 0 || !dims3) return SESSD_EINVAL;
  if (n_cap == 0) return SESSD_OK;
  const size_t total = (size_t)n_cap * channels;
  SESSD_LAUNCH((sparse_dense_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, features, indices, n_cap,
               n_dev, channels, dims3[0], dims3[1], dims3[2], const_cast<float*>(dense));
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_dense_to_sparse(const float* dense, const int* indices, int n, int channels, const int* dims3, float* features,
                          hipStream_t stream) {
  return sessd_dense_to_sparse_dev(dense, indices, n, nullptr, channels, dims3, features, stream);
}

}  // extern "C"
