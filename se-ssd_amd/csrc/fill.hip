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

// Several clears in ONE launch: the workgroups are dealt out over the segments in proportion to their sizes (a frame's
// launches cost ~5 us each on the device whatever their size: three clears of 0.5 / 3.5 / 18 MB as one launch).
struct FillSegs {
  uint32_t* p[4];
  uint32_t v[4];
  unsigned long long n[4];   // words
  unsigned blk_end[4];       // exclusive prefix of the segments' workgroup counts
};
__global__ __launch_bounds__(256) void fill_multi_kernel(FillSegs S) {
  int s = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (blockIdx.x >= S.blk_end[q]) s = q + 1;
  uint32_t* p = S.p[0];
  uint32_t v = S.v[0];
  unsigned long long n = S.n[0];
  unsigned b0 = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q == s) { p = S.p[q]; v = S.v[q]; n = S.n[q]; b0 = S.blk_end[q - 1]; }
  // 16 words (64 B) per thread: four 16-byte stores, consecutive threads on consecutive 16-byte pieces
  const size_t base = (size_t)(blockIdx.x - b0) * 4096;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const size_t i = base + (size_t)e * 1024 + (size_t)threadIdx.x * 4;
    if (i + 4 <= n) {
      *reinterpret_cast<uint4*>(p + i) = make_uint4(v, v, v, v);
    } else {
      for (size_t k = i; k < n; ++k) p[k] = v;
    }
  }
}
}  // namespace

// Up to four (pointer, value, word count) clears in one launch; every pointer 16-byte aligned.
extern "C" int sessd_fill_u32_multi(int n_segments, void* const* ptrs, const uint32_t* values, const size_t* n_words, hipStream_t stream) {
  if (n_segments < 1 || n_segments > 4 || !ptrs || !values || !n_words) return SESSD_EINVAL;
  FillSegs S;
  unsigned blk = 0;
  for (int q = 0; q < 4; ++q) {
    const bool on = q < n_segments;
    if (on && (!ptrs[q] || ((uintptr_t)ptrs[q] & 15))) return SESSD_EINVAL;
    S.p[q] = on ? (uint32_t*)ptrs[q] : nullptr;
    S.v[q] = on ? values[q] : 0u;
    S.n[q] = on ? (unsigned long long)n_words[q] : 0ull;
    if (on) {
      const size_t nb = (n_words[q] + 4095) / 4096;
      if (nb > 0x3fffffffu - blk) return SESSD_EINVAL;
      blk += (unsigned)nb;
    }
    S.blk_end[q] = blk;
  }
  if (blk == 0) return SESSD_OK;
  SESSD_LAUNCH(fill_multi_kernel, dim3(blk), dim3(256), 0, stream, S);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

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
