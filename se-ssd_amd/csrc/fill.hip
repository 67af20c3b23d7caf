// 32-bit pattern fill as a KERNEL (not hipMemsetAsync): every clear of the hot path goes through here.
// hipMemsetAsync nodes captured into a hipGraph were observed not to take effect on replay on this stack
// (hash tables kept stale keys from earlier frames until the open-addressing probe never terminated), so the
// library issues no memset at all; a kernel node has ordinary stream / graph ordering.
#include "common.hpp"

namespace {
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) {
    *reinterpret_cast<uint4*>(p + i) = make_uint4(v, v, v, v);
  } else {
    for (; i < n; ++i) p[i] = v;
  }
}
}  // namespace

// p must be 16-byte aligned; n = number of 32-bit words
int sessd_fill_u32_launch(void* p, uint32_t value, size_t n_words, hipStream_t stream) {
  if (n_words == 0) return SESSD_OK;
  const size_t threads = (n_words + 3) / 4;
  SESSD_LAUNCH(fill_u32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (uint32_t*)p, value,
                     n_words);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// When the caller clears the scratch of several stages with ONE fill over a contiguous arena (the inference
// engine does), the per-call clears inside sessd_voxelize_frame / sessd_sparse_downsample_sites are skipped.
// (per host thread: an engine enqueueing on one thread does not change these entry points for callers on other threads)
static thread_local int g_external_clear = 0;
int sessd_external_clear_enabled() { return g_external_clear; }
extern "C" void sessd_set_external_clear(int on) { g_external_clear = on ? 1 : 0; }

extern "C" int sessd_fill_u32(void* p, uint32_t value, size_t n_words, hipStream_t stream) {
  return sessd_fill_u32_launch(p, value, n_words, stream);
}
