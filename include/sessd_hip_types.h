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

#endif
