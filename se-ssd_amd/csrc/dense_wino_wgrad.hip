// Weight gradient of the 3x3 stride-1 BEV convolutions IN THE WINOGRAD DOMAIN (F(2x2,3x3)) on the f32 matrix cores: the
// backward of det3d/models/necks/rpn_v1.py:135-210 inside the training step (trainer_sessd.py:250-275), seven layers per
// iteration. The direct kernel (dense_grad.hip) multiplies 9 taps per (output pixel, co, ci): 41.5 GFLOP per layer at batch 4,
// 417 us = 99 TFLOP/s, the second largest item of the iteration. With Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile,
//
//   dU_xi[co][ci] = sum over tiles  dM_xi[co][tile] * V_xi[ci][tile]      xi = 0..15,  dM = A dY A^T,  V = B^T d B
//   dg[co][ci]    = G^T dU[co][ci] G                                        (3x3 from 4x4)
//
// i.e. 16 GEMMs with the TILES as reduction axis: 16 products per tile instead of 36 (18.5 GFLOP executed).
//
//   workgroup = 8 waves; a 64 co x 64 ci block, all 16 xi, one chunk of the tile list (split-K; partial dU per chunk, summed in
//               chunk order by the reduce kernel, which also applies G^T . G: deterministic, no float atomics)
//   stage     = 8 consecutive tiles. Thread (c = tid / 8, t = tid % 8) loads the 4x4 input patch of channel ci0 + c and the
//               2x2 output-gradient tile of channel co0 + c at tile t (16-byte / 8-byte buffer loads, image border by
//               out-of-range offsets and two lane masks, as in dense_wino_sk.hip), transforms both in registers and writes
//               dM[xi][c][t], V[xi][c][t] of the NEXT stage into the other LDS buffer (2 x 64 KB);
//   wave w    owns xi = 2w, 2w + 1 for the whole 64 x 64 block: 8 accumulators of 32x32. Its operands come from LDS as
//               ds_read_b128: lane (i, h) takes tiles 4h .. 4h+3 of channel i -- MFMA step e contracts the tile pair
//               {e, 4 + e} -- 8 reads for 32 MFMAs per stage, conflict-free. One barrier per stage.
// Loads for stage s + 1 are issued before the MFMAs of stage s. Numerics: float32 transforms and accumulation (Winograd
// rounding ~1e-6 of the result scale, like the forward kernels); the bits depend on (shape, number of chunks), not on timing.
#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using f32x2v = __attribute__((ext_vector_type(2))) float;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
#define SESSD_OOB 0x80000000u

constexpr int WW_NT = 512;
constexpr int WW_TILES = 8;                       // tiles per stage
constexpr int WW_OPER = 16 * 64 * WW_TILES;       // floats of one operand image [xi 16][c 64][t 8]
constexpr int WW_MAX_CHUNKS = 64;

struct WwArgs {
  const float* inp;    // (B, ci, H, W)
  const float* gout;   // (B, co, H, W)
  float* partial;      // [nchunks][16][co][ci]
  int B, ci, co, H, W;
  int tw, th, total_tiles, stages, nchunks, cob_n, cib_n;
};

__global__ __launch_bounds__(WW_NT, 1) void conv3x3s1_wino_wgrad_kernel(WwArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * WW_OPER];   // [buffer 2][dM | V][xi][c][t]: 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pairs = A.cob_n * A.cib_n;
  int chunk, pair;
  if ((A.nchunks & 7) == 0) {   // the (co, ci) blocks of one chunk on one XCD (workgroup b runs on XCD b % 8): they share its tiles in L2
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    pair = r % pairs;
    chunk = xcd + 8 * (r / pairs);
  } else {
    chunk = blockIdx.x / pairs;
    pair = blockIdx.x - chunk * pairs;
  }
  const int cob = pair / A.cib_n, cib = pair - cob * A.cib_n;
  const int s0 = (int)((long long)chunk * A.stages / A.nchunks), s1 = (int)((long long)(chunk + 1) * A.stages / A.nchunks);

  // ---- transform role: thread = (channel c of the 64-blocks, tile t of the stage)
  const int t8 = tid & 7, c = tid >> 3;
  const int plane = A.H * A.W;
  const rsrc_t xr = make_rsrc(A.inp, (unsigned)((size_t)A.B * A.ci * plane * 4));
  const rsrc_t gr = make_rsrc(A.gout, (unsigned)((size_t)A.B * A.co * plane * 4));
  int T = s0 * WW_TILES + t8;   // linear tile (b, ty, tx)
  int tx, ty, ex, eg;           // ex / eg: element index of the patch's top-left corner (row 2ty-1, column 2tx-1) / of the gout tile
  {
    const int b = T / (A.th * A.tw);
    ty = (T - b * A.th * A.tw) / A.tw;
    tx = T - (b * A.th + ty) * A.tw;
    ex = (b * A.ci + cib * 64 + c) * plane + (2 * ty - 1) * A.W + 2 * tx - 1;
    eg = (b * A.co + cob * 64 + c) * plane + 2 * ty * A.W + 2 * tx;
  }
  const int img_x = (A.ci - 1) * plane, img_g = (A.co - 1) * plane;

  f32x4v pr[4];
  f32x2v gy[2];
  bool mask_l = false, mask_r = false;
  // The 16-byte row segment of a patch: columns 2tx-1 .. 2tx+2; at the left border it starts one column later, at the right
  // border one column earlier (a load never crosses the end of a row, hence never the end of the tensor) and the lane masks of
  // the transform shift it back. Rows -1 and H get an out-of-range offset (the hardware returns zeros). Then the thread moves
  // 8 tiles on: +16 columns; past the end of a tile row +W more; past the end of an image + (channels - 1) planes (tw >= 8).
#define SESSD_WW_LOAD()                                                                             \
  {                                                                                                 \
    const bool live = T < A.total_tiles;                                                            \
    mask_l = live && tx == 0;                                                                       \
    mask_r = live && tx == A.tw - 1;                                                                \
    const int xs = ex + (tx == 0 ? 1 : 0) - (tx == A.tw - 1 ? 1 : 0);                               \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
      const bool row_ok = live && (q != 0 || ty > 0) && (q != 3 || ty < A.th - 1);                  \
      const unsigned off = row_ok ? (unsigned)(xs + q * A.W) * 4u : SESSD_OOB;                      \
      pr[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0)); \
    }                                                                                               \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                 \
      const unsigned off = live ? (unsigned)(eg + a * A.W) * 4u : SESSD_OOB;                        \
      gy[a] = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(gr, (int)off, 0, 0)); \
    }                                                                                               \
    T += WW_TILES; tx += WW_TILES; ex += 2 * WW_TILES; eg += 2 * WW_TILES;                          \
    if (tx >= A.tw) {                                                                               \
      tx -= A.tw; ++ty; ex += A.W; eg += A.W;                                                       \
      if (ty >= A.th) { ty -= A.th; ex += img_x; eg += img_g; }                                     \
    }                                                                                               \
  }
  // both transforms of the loaded tile into LDS buffer VOFF (floats): dM at +0, V at + WW_OPER
#define SESSD_WW_TRANSFORM(VOFF)                                                                    \
  {                                                                                                 \
    if (__builtin_amdgcn_ballot_w64(mask_l || mask_r) != 0) {                                       \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                               \
        const f32x4v p = pr[q];                                                                     \
        pr[q].x = mask_l ? 0.f : (mask_r ? p.y : p.x); pr[q].y = mask_l ? p.x : (mask_r ? p.z : p.y); \
        pr[q].z = mask_l ? p.y : (mask_r ? p.w : p.z); pr[q].w = mask_l ? p.z : (mask_r ? 0.f : p.w); \
      }                                                                                             \
    }                                                                                               \
    float* dst = &lds[(VOFF) + tid];   /* [xi][c][t]: c * 8 + t == tid */                           \
    {                                                                                               \
      const f32x2v r0 = gy[0], r1 = gy[0] + gy[1], r2 = gy[0] - gy[1], r3 = -gy[1];                 \
      const f32x2v rows[4] = {r0, r1, r2, r3};                                                      \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                               \
        dst[(a * 4 + 0) * 512] = rows[a].x;                                                         \
        dst[(a * 4 + 1) * 512] = rows[a].x + rows[a].y;                                             \
        dst[(a * 4 + 2) * 512] = rows[a].x - rows[a].y;                                             \
        dst[(a * 4 + 3) * 512] = -rows[a].y;                                                        \
      }                                                                                             \
    }                                                                                               \
    {                                                                                               \
      f32x2v tl[4], tr[4];                                                                          \
      tl[0] = pr[0].xy - pr[2].xy; tr[0] = pr[0].zw - pr[2].zw;                                     \
      tl[1] = pr[1].xy + pr[2].xy; tr[1] = pr[1].zw + pr[2].zw;                                     \
      tl[2] = pr[2].xy - pr[1].xy; tr[2] = pr[2].zw - pr[1].zw;                                     \
      tl[3] = pr[1].xy - pr[3].xy; tr[3] = pr[1].zw - pr[3].zw;                                     \
      float* dv = dst + WW_OPER;                                                                    \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                               \
        dv[(a * 4 + 0) * 512] = tl[a].x - tr[a].x;                                                  \
        dv[(a * 4 + 1) * 512] = tl[a].y + tr[a].x;                                                  \
        dv[(a * 4 + 2) * 512] = tr[a].x - tl[a].y;                                                  \
        dv[(a * 4 + 3) * 512] = tl[a].y - tr[a].y;                                                  \
      }                                                                                             \
    }                                                                                               \
  }

  f32x16 acc[2][2][2];   // [xi of the wave][co half][ci half]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][m][n][e] = 0.f;

  if (s0 < s1) {
    SESSD_WW_LOAD()
    SESSD_WW_TRANSFORM(0)
    __syncthreads();
    int voff = 0;
    // operand role: lane (i, h): channel i of a 32-block, tiles 4h .. 4h + 3 -- one 16-byte LDS read (indices in float4 units)
    const int i = lane & 31, h = lane >> 5;
    const f32x4v* lds4 = reinterpret_cast<const f32x4v*>(lds);
    const int ooff4 = wave * 256 + i * 2 + h;
    for (int s = s0; s < s1; ++s) {
      if (s + 1 < s1) SESSD_WW_LOAD()
      const int voff4 = voff >> 2;
      f32x4v av[2][2], bv[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          av[x][m] = lds4[voff4 + ooff4 + x * 128 + m * 64];
          bv[x][m] = lds4[voff4 + WW_OPER / 4 + ooff4 + x * 128 + m * 64];
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[x][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][m][e], bv[x][n][e], acc[x][m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < s1) SESSD_WW_TRANSFORM(voff ^ (2 * WW_OPER))
      __syncthreads();
      voff ^= 2 * WW_OPER;
    }
  }
#undef SESSD_WW_LOAD
#undef SESSD_WW_TRANSFORM

  // ---- this chunk's partial dU: D layout column (ci) = lane & 31, row (co) = (e & 3) + 8 * (e >> 2) + 4 * h
  {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int xi = wave * 2 + x;
      float* dst = A.partial + ((size_t)chunk * 16 + xi) * A.co * A.ci;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int co = cob * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int ci = cib * 64 + n * 32 + i;
            dst[(size_t)co * A.ci + ci] = acc[x][m][n][e];
          }
    }
  }
}

// thread = one (co, ci): dU_xi = sum over chunks (in chunk order), dg = G^T dU G
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ partial, int nchunks, int cc,
                                                                 float* __restrict__ gw) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= cc) return;
  float u[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) u[xi] = 0.f;
  for (int k = 0; k < nchunks; ++k) {
    float v[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) v[xi] = partial[((size_t)k * 16 + xi) * cc + p];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] += v[xi];
  }
  float t[3][4];   // G^T dU
#pragma unroll
  for (int col = 0; col < 4; ++col) {
    const float hs = 0.5f * (u[4 + col] + u[8 + col]), hd = 0.5f * (u[4 + col] - u[8 + col]);
    t[0][col] = u[col] + hs;
    t[1][col] = hd;
    t[2][col] = hs + u[12 + col];
  }
  float* o = gw + (size_t)p * 9;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float hs = 0.5f * (t[k][1] + t[k][2]), hd = 0.5f * (t[k][1] - t[k][2]);
    o[k * 3 + 0] = t[k][0] + hs;
    o[k * 3 + 1] = hd;
    o[k * 3 + 2] = hs + t[k][3];
  }
}

bool ww_shape_ok(int batch, int cin, int cout, int h, int w) {
  return batch >= 1 && cin >= 64 && cout >= 64 && cin % 64 == 0 && cout % 64 == 0 && h >= 2 && w >= 16 && (h & 1) == 0 && (w & 1) == 0 &&
         (size_t)batch * cin * h * w * 4 < 0x7FFFFFFFull && (size_t)batch * cout * h * w * 4 < 0x7FFFFFFFull;
}

int ww_chunks(int batch, int cin, int cout, int h, int w) {
  const int pairs = (cin / 64) * (cout / 64);
  const int stages = sessd_divup(batch * (h / 2) * (w / 2), WW_TILES);
  int n = 256 / pairs;                       // one workgroup per CU
  if (n > WW_MAX_CHUNKS) n = WW_MAX_CHUNKS;
  if (n > stages / 4) n = stages / 4;        // a chunk shorter than four stages is all pipeline fill
  if (n >= 8) n &= ~7;                       // multiples of 8: the XCD placement above
  return n < 1 ? 1 : n;
}

}  // namespace

extern "C" {

// 0: the shape is outside this kernel (use sessd_conv2d_wgrad)
size_t sessd_conv3x3_wgrad_winograd_workspace_bytes(int batch, int cin, int cout, int h, int w) {
  if (!ww_shape_ok(batch, cin, cout, h, w)) return 0;
  return (size_t)ww_chunks(batch, cin, cout, h, w) * 16 * cin * cout * sizeof(float);
}

// grad_weight (cout, cin, 3, 3) of Conv2d(cin, cout, 3, stride 1, padding 1): input (B, cin, h, w), grad_out (B, cout, h, w);
// cin, cout multiples of 64, h, w even, w >= 16. Same result as sessd_conv2d_wgrad(..., 3, 1) up to Winograd rounding.
int sessd_conv3x3_wgrad_winograd(const float* input, int batch, int cin, int h, int w, const float* grad_out, int cout,
                                 float* grad_weight, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!ww_shape_ok(batch, cin, cout, h, w) || !input || !grad_out || !grad_weight) return SESSD_EINVAL;
  if (workspace_bytes < sessd_conv3x3_wgrad_winograd_workspace_bytes(batch, cin, cout, h, w)) return SESSD_EWORKSPACE;
  WwArgs A;
  A.inp = input; A.gout = grad_out; A.partial = (float*)workspace;
  A.B = batch; A.ci = cin; A.co = cout; A.H = h; A.W = w;
  A.tw = w / 2; A.th = h / 2;
  A.total_tiles = batch * A.th * A.tw;
  A.stages = sessd_divup(A.total_tiles, WW_TILES);
  A.nchunks = ww_chunks(batch, cin, cout, h, w);
  A.cob_n = cout / 64; A.cib_n = cin / 64;
  SESSD_LAUNCH(conv3x3s1_wino_wgrad_kernel, dim3(A.nchunks * A.cob_n * A.cib_n), dim3(WW_NT), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  const int cc = cout * cin;
  SESSD_LAUNCH(wino_wgrad_reduce_kernel, dim3(sessd_divup(cc, 256)), dim3(256), 0, stream, (const float*)workspace, A.nchunks, cc,
               grad_weight);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
