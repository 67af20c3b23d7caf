// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 / 16x16x4 rate, NACC independent accumulators per wave,
// WPS waves per SIMD. Used to calibrate what fraction of the 157.3 TFLOP/s datasheet peak is reachable.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// Same with operands that differ per lane and per instruction (pseudo-random in [-1, 1)): with constant operands the matrix
// pipe's inputs never toggle, the chip draws less power and holds a higher clock than any real GEMM sees.
template <int NACC>
__global__ __launch_bounds__(256) void k32r(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float av[8], bv[8];
  unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u; av[i] = (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f + a0 * 1e-9f;
    s = s * 1664525u + 1013904223u; bv[i] = (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f + b0 * 1e-9f;
  }
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + 3 * i) & 7], acc[i], 0, 0, 0);
  }
  float t = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <typename K>
void run(const char* name, K kern, int nacc, int blocks, double flop_per_mfma) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfma = (double)blocks * 4 * iters * 4 * nacc;
  printf("%s nacc=%d blocks=%d: %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per wave)\n", name, nacc, blocks, ms, mfma * flop_per_mfma / ms / 1e9, ms * 1e6 / (iters * 4.0 * nacc) / ((blocks + 255) / 256));
  hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 1024}) {
    run("32x32x2", k32<1>, 1, blocks, 4096.0);
    run("32x32x2", k32<2>, 2, blocks, 4096.0);
    run("32x32x2", k32<4>, 4, blocks, 4096.0);
    run("32x32x2 random operands", k32r<4>, 4, blocks, 4096.0);
    run("16x16x4", k16<2>, 2, blocks, 2048.0);
    run("16x16x4", k16<4>, 4, blocks, 2048.0);
  }
  return 0;
}
