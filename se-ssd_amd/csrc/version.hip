// Library identification (checked by the Python loader and the symbol-export test).
#include "common.hpp"
extern "C" const char* sessd_version(void) { return "sessd_hip 0.1 gfx950"; }

// A HIP stream whose kernels are confined to a subset of the compute units (hipExtStreamCreateWithCUMask): mask = n_words 32-bit
// words, bit i = CU i of the device's numbering. Used to give each of several frames in flight CUs of its own: the persistent
// stream-K launches of one frame then never hold the other frame's small kernels back (round 5, bench.py --cu-split).
extern "C" int sessd_stream_create_cu_mask(int n_words, const uint32_t* mask, hipStream_t* stream) {
  if (n_words < 1 || !mask || !stream) return SESSD_EINVAL;
  SESSD_TRY(hipExtStreamCreateWithCUMask(stream, (uint32_t)n_words, mask));
  return SESSD_OK;
}
extern "C" int sessd_stream_destroy(hipStream_t stream) {
  SESSD_TRY(hipStreamDestroy(stream));
  return SESSD_OK;
}

// Which physical compute units a stream's kernels land on (test / diagnostics of the CU-masked streams above): `n_workgroups`
// workgroups of one wave each spin for ~`spin_cycles` shader cycles (so that the dispatcher spreads them over every CU the queue may
// use instead of re-using the first free one) and record ids[workgroup] = XCC_ID << 16 | (HW_ID & 0xFFFF): HW_ID bits 11:8 = CU,
// 12 = shader array, 15:13 = shader engine (gfx9 HW_REG_HW_ID), XCC_ID = the accelerator die. (xcc, se, sh, cu) names a physical CU.
namespace {
__global__ __launch_bounds__(64) void cu_probe_kernel(uint32_t* __restrict__ ids, int spin_cycles) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, bits 3:0
  const unsigned long long t0 = __builtin_readcyclecounter();
  while ((long long)(__builtin_readcyclecounter() - t0) < (long long)spin_cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) ids[blockIdx.x] = (xcc << 16) | (hw & 0xFFFFu);
}
}  // namespace
extern "C" int sessd_debug_cu_probe(uint32_t* ids, int n_workgroups, int spin_cycles, hipStream_t stream) {
  if (!ids || n_workgroups < 1 || spin_cycles < 0) return SESSD_EINVAL;
  SESSD_LAUNCH(cu_probe_kernel, dim3(n_workgroups), dim3(64), 0, stream, ids, spin_cycles);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}
