// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the SE-SSD hot path.
// Wave width is 64 on CDNA4; every wave-level idiom below is written for that.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SESSD_WAVE 64

// Every C-ABI entry point returns 0 on success or a hipError_t / negative
// argument-error code; nothing in this library calls exit() (the reference's
// iou3d.cpp:13-21 printf+exit behaviour is deliberately not reproduced).
#define SESSD_OK 0
#define SESSD_EINVAL (-1)
#define SESSD_EWORKSPACE (-2)

// hipGetLastError() is sticky per thread: an error left behind by ANOTHER library's runtime call (observed: 100 from
// torch's lazy device initialisation) would be reported by the next SESSD_CHECK_LAUNCH. Every launch clears it first.
#define SESSD_LAUNCH(...)              \
  do {                                  \
    (void)hipGetLastError();            \
    hipLaunchKernelGGL(__VA_ARGS__);    \
  } while (0)

#define SESSD_CHECK_LAUNCH()                          \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)

#define SESSD_TRY(expr)                               \
  do {                                                \
    hipError_t e__ = (expr);                          \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)

// kernel-based clear (fill.hip); the library never calls hipMemsetAsync (see fill.hip)
int sessd_fill_u32_launch(void* p, uint32_t value, size_t n_words, hipStream_t stream);
#define SESSD_FILL(ptr, value, n_words, stream)                                        \
  do {                                                                                 \
    int rc__ = sessd_fill_u32_launch((void*)(ptr), (uint32_t)(value), (size_t)(n_words), stream); \
    if (rc__ != 0) return rc__;                                                        \
  } while (0)

// scratch clears that an engine may take over with one arena-wide fill (sessd_set_external_clear)
int sessd_external_clear_enabled();
#define SESSD_FILL_SCRATCH(ptr, value, n_words, stream)                 \
  do {                                                                   \
    if (!sessd_external_clear_enabled()) SESSD_FILL(ptr, value, n_words, stream); \
  } while (0)

static inline __host__ __device__ int sessd_divup(int a, int b) { return (a + b - 1) / b; }
static inline __host__ __device__ size_t sessd_align(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- open-addressing hash: linear cell key -> row ------------------------
// EMPTY key is 0x7F7F7F7F so that one hipMemsetAsync(0x7F) clears keys,
// values and the voxelizer's per-entry index lists in a single pass.
#define SESSD_HASH_EMPTY 0x7F7F7F7Fu
#define SESSD_SENT 0x7F7F7F7F  // "no index" sentinel for int lists (same byte pattern)

static __device__ __forceinline__ uint32_t sessd_hash_u32(uint32_t k) {
  k ^= k >> 16;
  k *= 0x7feb352du;
  k ^= k >> 15;
  k *= 0x846ca68bu;
  k ^= k >> 16;
  return k;
}

// Home slot and probe sequence (round 4). The 8 cells of an aligned run along x (the fastest axis of the linear cell key) share
// one 32-byte BUCKET: slot = bucket(hash(key >> 3)) * 8 + (key & 7). A rulebook looks up x-1, x, x+1 of nine (z, y) rows per
// site: with a slot per hashed cell those were 27 random cache lines (dense-scene batch: 2.05 GB fetched by one
// chain_rulebook launch, profiles/r4_sparse_pmc.txt), now the three of a row are one line. A taken slot sends the key to the SAME
// offset of the next bucket (so a displaced run stays a run); after a whole lap of that offset class the sequence moves to the
// next offset: every slot of the table is visited once, insert and find walk the same sequence.
static __device__ __forceinline__ uint32_t sessd_hash_home(uint32_t key, uint32_t mask) {
  return ((sessd_hash_u32(key >> 3) << 3) | (key & 7u)) & mask;
}
#define SESSD_HASH_ADVANCE(slot, lap, mask)  \
  {                                          \
    (slot) = ((slot) + 8u) & (mask);         \
    if ((slot) == (lap)) {                   \
      (slot) = ((slot) + 1u) & (mask);       \
      (lap) = (slot);                        \
    }                                        \
  }

// Insert-or-find. Returns the slot that holds `key`, or SESSD_HASH_FULL when every slot is taken by other keys
// (the probe sequence is bounded by the table size: a full table is reported, never spun on).
#define SESSD_HASH_FULL 0xFFFFFFFFu
static __device__ __forceinline__ uint32_t sessd_hash_insert(uint32_t* keys, uint32_t mask, uint32_t key) {
  uint32_t slot = sessd_hash_home(key, mask), lap = slot;
  for (uint32_t probes = 0; probes <= mask; ++probes) {
    uint32_t prev = atomicCAS(&keys[slot], SESSD_HASH_EMPTY, key);
    if (prev == SESSD_HASH_EMPTY || prev == key) return slot;
    SESSD_HASH_ADVANCE(slot, lap, mask)
  }
  return SESSD_HASH_FULL;
}

// Lookup. Returns value or -1.
static __device__ __forceinline__ int sessd_hash_find(const uint32_t* __restrict__ keys,
                                                      const int* __restrict__ vals, uint32_t mask,
                                                      uint32_t key) {
  uint32_t slot = sessd_hash_home(key, mask), lap = slot;
  for (uint32_t probes = 0; probes <= mask; ++probes) {
    uint32_t k = keys[slot];
    if (k == key) {
      int v = vals[slot];
      return v == SESSD_SENT ? -1 : v;
    }
    if (k == SESSD_HASH_EMPTY) return -1;
    SESSD_HASH_ADVANCE(slot, lap, mask)
  }
  return -1;  // full table without the key
}

// ---- wave / block reductions ----------------------------------------------
static __device__ __forceinline__ int sessd_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Exclusive scan of one int per thread across a block of NT threads (NT % 64 == 0,
// NT <= 1024). `smem` must hold NT/64 ints. Returns the exclusive prefix; *total
// receives the block sum.
template <int NT>
static __device__ __forceinline__ int sessd_block_exscan(int v, int* smem, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) smem[wid] = incl;
  __syncthreads();
  int wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    int s = smem[w];
    if (w < wid) wbase += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wbase + incl - v;
}
