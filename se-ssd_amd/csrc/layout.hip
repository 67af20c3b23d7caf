// The head outputs' layout change of det3d/models/bbox_heads/mg_head_sessd.py:217-230 (each 1x1 conv's output
// `.permute(0, 2, 3, 1).contiguous()`): the fused 22-channel head tensor (B, C, H*W) split into its parts in NHWC, ONE launch
// instead of a strided copy per part, and the adjoint (the parts' gradients back into one planar tensor) for the training step.
// Pure data movement: thread = one pixel, reads coalesced along the plane, writes each part's channels of its pixel.
#include "common.hpp"

namespace {

struct SplitArgs {
  float* part[4];      // (B, plane, size[k]) each; may be NULL in the merge direction (= a zero gradient)
  int size[4];
  int start[4];        // first planar channel of part k
  int n;
};

template <bool MERGE>
__global__ __launch_bounds__(256) void nchw_split_nhwc_kernel(float* __restrict__ planar, SplitArgs A, int B, int C, int plane) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * plane) return;
  const int b = (int)(i / plane), p = (int)(i - (long long)b * plane);
  float* src = planar + (size_t)b * C * plane + p;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= A.n) break;
    float* o = A.part[k] ? A.part[k] + (size_t)i * A.size[k] : nullptr;
    for (int c = 0; c < A.size[k]; ++c) {
      float* s = src + (size_t)(A.start[k] + c) * plane;
      if (MERGE)
        *s = o ? o[c] : 0.f;
      else
        o[c] = *s;
    }
  }
}

int fill_args(SplitArgs& A, int channels, int n_parts, const int* sizes, float* const* parts, bool need_all) {
  if (n_parts < 1 || n_parts > 4 || !sizes || !parts) return SESSD_EINVAL;
  int at = 0;
  A.n = n_parts;
  for (int k = 0; k < 4; ++k) {
    A.part[k] = k < n_parts ? parts[k] : nullptr;
    A.size[k] = k < n_parts ? sizes[k] : 0;
    A.start[k] = at;
    if (k < n_parts) {
      if (sizes[k] < 1 || (need_all && !parts[k])) return SESSD_EINVAL;
      at += sizes[k];
    }
  }
  return at == channels ? SESSD_OK : SESSD_EINVAL;
}

}  // namespace

extern "C" {

// planar (batch, channels, plane) -> parts[k] (batch, plane, sizes[k]), k < n_parts <= 4, sum of sizes == channels; HOST arrays
// `sizes` and `parts` (device pointers).
int sessd_nchw_split_nhwc(const float* planar, int batch, int channels, int plane, int n_parts, const int* sizes, float* const* parts,
                          hipStream_t stream) {
  SplitArgs A;
  if (!planar || batch < 1 || channels < 1 || plane < 1) return SESSD_EINVAL;
  const int rc = fill_args(A, channels, n_parts, sizes, parts, true);
  if (rc != SESSD_OK) return rc;
  const long long n = (long long)batch * plane;
  SESSD_LAUNCH((nchw_split_nhwc_kernel<false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, const_cast<float*>(planar), A,
               batch, channels, plane);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// The adjoint: parts' gradients (a NULL part = zeros) -> planar gradient (batch, channels, plane), every element written.
int sessd_nhwc_merge_nchw(float* const* parts, int n_parts, const int* sizes, int batch, int channels, int plane, float* planar,
                          hipStream_t stream) {
  SplitArgs A;
  if (!planar || batch < 1 || channels < 1 || plane < 1) return SESSD_EINVAL;
  const int rc = fill_args(A, channels, n_parts, sizes, parts, false);
  if (rc != SESSD_OK) return rc;
  const long long n = (long long)batch * plane;
  SESSD_LAUNCH((nchw_split_nhwc_kernel<true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, planar, A, batch, channels, plane);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
