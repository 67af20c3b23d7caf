/* Plain-C argument structs of libsessd_hip.so (included by sessd_hip.h and by the kernels' sources). */
#ifndef SESSD_HIP_TYPES_H
#define SESSD_HIP_TYPES_H

#include <stdint.h>

/* One strided SparseConv3d step of a chain of sparse levels: level l-1 -> level l (level 0 = the voxels). */
typedef struct {
  int32_t ksize[3], stride[3], pad[3]; /* (z, y, x); ksize >= stride */
  int32_t out_dims[3];                 /* spatial shape (D, H, W) of level l */
  int32_t cap;                         /* row capacity of level l */
  int32_t* indices;                    /* out: (cap, 4) int32 [b, z, y, x], rows in ascending (b, z, y, x) order */
  int32_t* n_dev;                      /* out: live rows of level l (device int) */
} sessd_chain_level_t;

/* One neighbour table ("rulebook") over the output sites of `out_level`, looked up among the sites of `in_level`
 * (submanifold conv: in_level == out_level, stride 1, pad = ksize / 2). */
typedef struct {
  int32_t in_level, out_level;
  int32_t ksize[3], stride[3], pad[3];
  int32_t* nbr;        /* out: (kernel_volume, cap of out_level) input row or -1 */
  uint32_t* tile_mask; /* out: ceil(cap / 16) words, bit k = some site of the 16-site tile has a neighbour at offset k */
} sessd_rulebook_job_t;

/* One weight packing of a batched re-pack launch (sessd_dense_pack_batch): the arguments of sessd_conv2d_pack_taps (kind 0) or
 * sessd_conv3x3_winograd_pack (kind 1) plus the first block of the job inside the launch (256 threads per block; a job takes
 * ceil(elements / 256) blocks, elements = (cin/2) * ntaps * 2 * cout_pad32 for kind 0, cout_pad * cin for kind 1). */
typedef struct {
  const float* w;
  float* out;
  long long out_stride, in_stride;
  int32_t tap_off[16];
  int32_t cout, cin, ntaps;
  int32_t kind, flip, layout;
  int32_t block_start, pad_;
} sessd_dense_pack_job_t;

/* One sparse-conv weight packing of sessd_sparse_pack_batch: sessd_sparse_pack_weight (adjoint 0) or
 * sessd_sparse_pack_weight_adjoint (adjoint 1; reverse_k = its reverse_offsets); cin / cout are those of the STORED weight
 * (kernel_volume, cin, cout); a job takes ceil(kernel_volume * cin * cout / 256) blocks. */
typedef struct {
  const float* w;
  float* out;
  int32_t kernel_volume, cin, cout, adjoint, reverse_k, block_start;
} sessd_sparse_pack_job_t;

#endif
