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
  /* optional (all three or none): OFFSET-PATTERN TILES. site_mask: out, (cap) words, bit k = site has a neighbour at offset k.
   * perm: out, ceil(cap / 256) * 256 bytes -- inside every group of 256 consecutive rows the live sites sorted by site_mask;
   * position p of the group holds row  group * 256 + perm[group * 256 + p]. tile_mask_sorted: out, ceil(cap / 256) * 16 words,
   * the tile masks of the 16-position tiles of that order. sessd_sparse_conv_sorted walks these tiles: sites with the same
   * neighbour pattern share a tile, so fewer (tile, offset) steps multiply rows without a neighbour -- same results per site. */
  uint32_t* site_mask;
  uint8_t* perm;
  uint32_t* tile_mask_sorted;
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

/* One layer of sessd_fill_inactive_tiles: out (batch, cout, h, w), value[cout], tile_mask[batch][mask_th][2] 64-bit words (bit tx of
 * a row's 128 bits set = computed tile; mask_th >= h/2 rows per image), as sessd_bev_tile_activity writes them. */
typedef struct {
  float* out;
  const float* value;
  const uint64_t* tile_mask;
  int32_t cout, h, w, mask_th;
  int32_t tile;   /* 0 / 2: 2x2-pixel tiles, value[cout]; 4: 4x4-pixel tiles (a transposed conv over 2x2 tiles of its input),
                     value[4][cout] = the constant of each output parity class (py * 2 + px); 6: 2x2-pixel tiles with such a
                     value[4][cout] table (a 3x3 layer behind the transposed convs: its input is constant per parity class) */
  int32_t near_kind;           /* how near_mask's reader reaches this map: 0 = a 3x3 stride-1 layer on the SAME tile grid (reach:
                                  the 3x3 tiles around a listed tile); on a grid TWICE AS COARSE: 1 = it touches the map inside its
                                  listed tiles only (a residual), 2 = a 3x3 stride-2 layer over 2x2 tiles of its output (reach:
                                  tiles 2Ty-1 .. 2Ty+1, 2Tx-1 .. 2Tx+1 of this map) */
  const uint64_t* near_mask;   /* NULL, or (tile 2 only) the tile mask of the list-driven reader(s) of `out`: only the tiles within
                                  its reach are filled, the others keep what they hold */
} sessd_fill_tiles_job_t;

/* One sparse-conv weight packing of sessd_sparse_pack_batch: sessd_sparse_pack_weight (adjoint 0) or
 * sessd_sparse_pack_weight_adjoint (adjoint 1; reverse_k = its reverse_offsets); cin / cout are those of the STORED weight
 * (kernel_volume, cin, cout); a job takes ceil(kernel_volume * cin * cout / 256) blocks. */
typedef struct {
  const float* w;
  float* out;
  int32_t kernel_volume, cin, cout, adjoint, reverse_k, block_start;
} sessd_sparse_pack_job_t;

/* The head outputs and targets of ONE network (student or EMA teacher) as sessd_head_loss reads them: A anchors per sample in
 * the reference's order (anchor = (y * W + x) * 2 + rotation), B samples. box / dir / iou / cls are the NHWC tensors that
 * MultiGroupHead.forward returns (mg_head_sessd.py:217-230), viewed as (B, A, 7) / (B, A, 2) / (B, A) / (B, A). */
typedef struct {
  const float* box;          /* (B, A, 7) box codes */
  const float* cls;          /* (B, A) class logits (one class) */
  const float* dir;          /* (B, A, 2) direction logits */
  const float* iou;          /* (B, A) IoU predictions */
  const void* labels;        /* (B, A) int32 or int64 (cfg.labels_i64): -1 ignore, 0 negative, > 0 positive */
  const float* reg_targets;  /* (B, A, 7) encoded regression targets */
  const float* anchors;      /* (B, A, 7) [x, y, z, w, l, h, r] */
} sessd_head_loss_net_t;

/* Shapes, capacities and the constants of examples/second/configs/config.py that sessd_head_loss needs. */
typedef struct {
  int32_t batch, num_anchors, labels_i64;
  int32_t pos_capacity;      /* positive anchors of the whole batch kept per network (more: flagged in record[48], dropped) */
  int32_t cons_capacity;     /* consistency candidates kept per sample and network (score >= score_thresh inside center_range) */
  float pos_cls_weight, neg_cls_weight;             /* loss_norm (NormByNumPositives) */
  float focal_alpha, focal_gamma, smooth_l1_sigma;  /* loss_cls, loss_bbox.sigma */
  float cls_loss_weight, loc_loss_weight, dir_loss_weight, direction_offset;
  float score_thresh, match_iou_thresh;             /* 0.3 (mg_head_sessd.py:653) and 0.7 (:573) */
  float center_range[6];                            /* post_center_range (mg_head_sessd.py:484) */
} sessd_head_loss_cfg_t;

#endif
