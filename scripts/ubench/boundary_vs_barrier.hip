// What does a dependency between two short phases cost on MI355X: a KERNEL BOUNDARY (the phases as two launches of one hipGraph)
// or a GRID BARRIER inside one persistent launch (release fence, agent-scope arrival counter, bounded spin, acquire fence)?
// Round-4 review item 4: DESIGN.md argued against a fused per-level launch for the submanifold layers of SpMiddleFHD (levels 3 - 4:
// <= 190 tiles, three dependent layers each) without measuring it. This measures the two primitives at that geometry (190 and
// 470 workgroups of 256 threads, every phase reads what ALL workgroups of the previous phase wrote -- the sparse convs gather
// neighbour rows from anywhere), alone and with a second stream keeping every CU busy (the two-frames-in-flight configuration).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bvb scripts/ubench/boundary_vs_barrier.hip && /tmp/bvb
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

// one phase: every thread reads a value another workgroup wrote in the previous phase, adds one, writes its own
__device__ __forceinline__ void phase_body(const int* __restrict__ in, int* __restrict__ out, int n_wg, int phase) {
  const int wg = blockIdx.x, t = threadIdx.x;
  const int src = (wg * 7 + 3 + phase) % n_wg;   // some other workgroup's row
  out[wg * 256 + t] = in[src * 256 + t] + 1;
}

__global__ __launch_bounds__(256) void phase_kernel(const int* in, int* out, int n_wg, int phase) { phase_body(in, out, n_wg, phase); }

// persistent form: `phases` phases separated by a monotonic-counter grid barrier. Every spin is bounded (err = 1 on a time-out).
__global__ __launch_bounds__(256) void fused_kernel(int* a, int* b, int n_wg, int phases, unsigned* counter, int* err) {
  for (int p = 0; p < phases; ++p) {
    phase_body((p & 1) ? b : a, (p & 1) ? a : b, n_wg, p);
    if (p + 1 == phases) break;
    // release: this workgroup's stores reach memory before it is counted
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(p + 1) * (unsigned)n_wg;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000) { *err = 1; break; }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

// a second stream's load: persistent workgroups that stream memory for `iters` rounds (never wait for anybody)
__global__ __launch_bounds__(512) void busy_kernel(const float4* src, float* sink, size_t n4, int iters) {
  float acc = 0.f;
  for (int it = 0; it < iters; ++it)
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 512) {
      const float4 v = src[i];
      acc += v.x + v.y + v.z + v.w;
    }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int PH = 48;
  hipStream_t s, s2;
  CHECK(hipStreamCreate(&s));
  CHECK(hipStreamCreate(&s2));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t busy_n4 = (size_t)64 << 20 >> 4;   // 64 MB
  float4* busy_src; float* sink;
  CHECK(hipMalloc(&busy_src, busy_n4 * 16));
  CHECK(hipMemset(busy_src, 0, busy_n4 * 16));
  CHECK(hipMalloc(&sink, 64));
  for (int n_wg : {190, 470, 1024}) {
    int *a, *b, *err; unsigned* counter;
    CHECK(hipMalloc(&a, n_wg * 256 * 4)); CHECK(hipMalloc(&b, n_wg * 256 * 4));
    CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&counter, 4));
    CHECK(hipMemset(a, 0, n_wg * 256 * 4)); CHECK(hipMemset(b, 0, n_wg * 256 * 4)); CHECK(hipMemset(err, 0, 4));
    // ---- the phases as PH launches of one graph
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int p = 0; p < PH; ++p)
      hipLaunchKernelGGL(phase_kernel, dim3(n_wg), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, n_wg, p);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    // one launch alone (its fixed cost is in both forms once)
    hipGraph_t g1; hipGraphExec_t ge1;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(phase_kernel, dim3(n_wg), dim3(256), 0, s, a, b, n_wg, 0);
    CHECK(hipStreamEndCapture(s, &g1));
    CHECK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
    hipGraph_t gf; hipGraphExec_t gef;
    const bool fits = n_wg <= 256 * 8;   // 256-thread workgroups: up to 8 per CU are resident
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    CHECK(hipMemsetAsync(counter, 0, 4, s));
    hipLaunchKernelGGL(fused_kernel, dim3(n_wg), dim3(256), 0, s, a, b, n_wg, PH, counter, err);
    CHECK(hipStreamEndCapture(s, &gf));
    CHECK(hipGraphInstantiate(&gef, gf, nullptr, nullptr, 0));
    for (int busy = 0; busy < 2; ++busy) {
      auto timed = [&](hipGraphExec_t x, int reps) {
        for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(x, s));
        CHECK(hipStreamSynchronize(s));
        if (busy) hipLaunchKernelGGL(busy_kernel, dim3(256), dim3(512), 0, s2, busy_src, sink, busy_n4, 40);
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(x, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipStreamSynchronize(s2));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
      };
      const float t_one = timed(ge1, 20), t_chain = timed(ge, 20);
      float t_fused = -1.f;
      if (fits) {
        // the counter memset is a graph node: on this stack a replayed memset node is unreliable -> clear it by hand each time
        float acc = 0;
        const int reps = 20;
        for (int i = 0; i < reps + 3; ++i) {
          CHECK(hipMemsetAsync(counter, 0, 4, s));
          if (i == 3 && busy) hipLaunchKernelGGL(busy_kernel, dim3(256), dim3(512), 0, s2, busy_src, sink, busy_n4, 40);
          CHECK(hipEventRecord(e0, s));
          hipLaunchKernelGGL(fused_kernel, dim3(n_wg), dim3(256), 0, s, a, b, n_wg, PH, counter, err);
          CHECK(hipEventRecord(e1, s));
          CHECK(hipStreamSynchronize(s));
          float ms = 0;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (i >= 3) acc += ms * 1e3f;
        }
        CHECK(hipStreamSynchronize(s2));
        t_fused = acc / reps;
      }
      int herr = 0;
      CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("workgroups %4d  other stream %s : one launch %.2f us | %d dependent launches %.2f us -> %.2f us per boundary+phase | "
             "one launch with %d grid barriers %.2f us -> %.2f us per barrier+phase%s\n",
             n_wg, busy ? "BUSY" : "idle", t_one, PH, t_chain, (t_chain - t_one) / (PH - 1), PH - 1, t_fused,
             t_fused > 0 ? (t_fused - t_one) / (PH - 1) : -1.f, herr ? "  [SPIN TIME-OUT]" : "");
    }
    CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(err)); CHECK(hipFree(counter));
  }
  return 0;
}
