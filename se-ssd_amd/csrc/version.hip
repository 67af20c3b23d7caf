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
