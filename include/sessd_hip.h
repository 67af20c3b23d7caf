/* libsessd_hip.so -- C ABI of the MI355X (gfx950) SE-SSD inference hot path.
 *
 * Plain C: raw DEVICE pointers, sizes and a hipStream_t (passed as void*). No torch types.
 * Every function returns 0 on success, a negative SESSD_E* code for argument errors, or a
 * positive hipError_t. Nothing here synchronises the host, allocates device memory, prints or
 * exits: workspaces are caller-provided (size queries below) and results stay on the device,
 * so a whole frame can be enqueued (or captured in a hipGraph) without a host round trip.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * Vegeta2020/SE-SSD tree). INTEGRATION.md shows the ctypes binding a maintainer would add.
 */
#ifndef SESSD_HIP_H
#define SESSD_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "sessd_hip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SESSD_OK 0
#define SESSD_EINVAL (-1)
#define SESSD_EWORKSPACE (-2)

typedef void* sessd_stream_t; /* hipStream_t */

const char* sessd_version(void);
/* 32-bit pattern fill as a kernel launch (graph-safe clear; the library itself never uses hipMemsetAsync) */
int sessd_fill_u32(void* ptr, uint32_t value, size_t n_words, sessd_stream_t stream);
/* up to four such clears (16-byte aligned pointers) in ONE launch: host arrays ptrs[n] / values[n] / n_words[n] */
int sessd_fill_u32_multi(int n_segments, void* const* ptrs, const uint32_t* values, const size_t* n_words,
                         sessd_stream_t stream);
/* on != 0: sessd_voxelize_frame / sessd_sparse_downsample_sites stop clearing their scratch (per-cell lists + cut word
 * at the start of the voxelizer workspace; output hash keys/vals and the first `out_hash_capacity` words of the
 * downsample workspace) -- the caller fills them with 0x7F7F7F7F itself, e.g. one fill over a contiguous arena.
 * The switch is per calling host thread (thread-local) and is read when an entry point is CALLED, not when its kernels run. */
void sessd_set_external_clear(int on);
/* A stream confined to a subset of the compute units (hipExtStreamCreateWithCUMask): mask[n_words], bit i = CU i. Several frames
 * in flight on disjoint CU sets do not hold each other's kernels back (bench.py --cu-split). */
int sessd_stream_create_cu_mask(int n_words, const uint32_t* mask, sessd_stream_t* stream);
int sessd_stream_destroy(sessd_stream_t stream);
/* Diagnostics of the CU-masked streams: n_workgroups one-wave workgroups spin ~spin_cycles shader cycles each and record
 * ids[workgroup] = XCC_ID << 16 | (HW_REG_HW_ID & 0xFFFF) -- bits 11:8 CU, 12 shader array, 15:13 shader engine: the physical CU the
 * workgroup ran on. tests/test_cu_mask_gpu.py: two streams of different CU sets never share a CU, also under hipGraph replay. */
int sessd_debug_cu_probe(uint32_t* ids, int n_workgroups, int spin_cycles, sessd_stream_t stream);

/* ------------------------------------------------------------------ voxelizer (a1-a3)
 * replaces det3d/ops/point_cloud/point_cloud_ops_v2.py:120-194 points_to_voxel (numba, CPU),
 *          det3d/core/input/voxel_generator.py:24-32 VoxelGenerator.generate,
 *          det3d/models/readers/voxel_encoder.py:215-220 VoxelFeatureExtractorV3.forward (mean_feat).
 * Bit-exact with the reference's serial first-come-first-served loop, including the `break`
 * at max_voxels and the <= max_points_per_voxel rule. The hash (keys/vals) afterwards maps
 * cell -> output row and is reused as the level-0 site index of SpMiddleFHD.
 * The slot layout of keys/vals is PRIVATE to the library (open addressing in 32-byte buckets: the 8 cells of an aligned run
 * along x share a bucket, csrc/common.hpp sessd_hash_home): a table is only ever filled by sessd_voxelize_frame(s) or
 * sessd_sparse_hash_build and read by the entry points that take (keys, vals, capacity); callers own the memory, not the layout. */
uint32_t sessd_hash_capacity(int max_items);
int sessd_hash_clear(uint32_t* keys, int32_t* vals, uint32_t capacity, sessd_stream_t stream);
size_t sessd_voxelize_workspace_bytes(uint32_t hash_capacity, int max_points_in_frame, int max_points_per_voxel,
                                      int max_voxels);
int sessd_voxelize_frame(const float* points, int num_points, int ndim, const float* range6, const float* voxel_size3,
                         const int32_t* grid3, int max_points_per_voxel, int max_voxels, int batch_index,
                         uint32_t* hash_keys, int32_t* hash_vals, uint32_t hash_capacity, float* voxels, int32_t* coors,
                         int coors_stride, int32_t* num_points_per_voxel, float* mean_feat, int32_t* prefix,
                         void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* ALL frames of a batch in FOUR launches (sessd_voxelize_frame: four per frame, frame after frame). points (batch,
 * points_per_frame, ndim): every frame holds exactly points_per_frame rows (pad with out-of-range rows: sessd_stage_points);
 * prefix[0] must be 0 on entry, prefix[1 .. batch] are written. Same outputs, bit for bit, as `batch` calls of
 * sessd_voxelize_frame with batch_index 0 .. batch - 1 on one shared hash (replaces the same reference loop,
 * point_cloud_ops_v2.py:9-62, once per frame of a collated batch: torchie/parallel/collate.py:154-218). */
size_t sessd_voxelize_frames_workspace_bytes(uint32_t hash_capacity, int batch, int points_per_frame, int max_points_per_voxel,
                                             int max_voxels);
int sessd_voxelize_frames(const float* points, int batch, int points_per_frame, int ndim, const float* range6,
                          const float* voxel_size3, const int32_t* grid3, int max_points_per_voxel, int max_voxels,
                          uint32_t* hash_keys, int32_t* hash_vals, uint32_t hash_capacity, float* voxels, int32_t* coors,
                          int coors_stride, int32_t* num_points_per_voxel, float* mean_feat, int32_t* prefix, void* workspace,
                          size_t workspace_bytes, sessd_stream_t stream);
/* (n,4) points -> fixed-capacity staging buffer, tail rows set out of range (dropped by the voxelizer) */
int sessd_stage_points(const float* points, int num_points, float* dst, int capacity, sessd_stream_t stream);
int sessd_vfe_mean(const float* voxels, const int32_t* num_points, const int32_t* num_voxels_dev, int num_voxels_host,
                   int max_points_per_voxel, int ndim, int num_features, float* out, sessd_stream_t stream);

/* ------------------------------------------------------------------ iou3d_cuda operators (a15)
 * replace det3d/core/iou3d/src/iou3d.cpp:24-115 (boxes_overlap_bev_gpu, boxes_aligned_overlap_bev_gpu,
 * boxes_iou_bev_gpu, boxes_iou3d_gpu) and :117-262 (nms_gpu, nms_3d_gpu, nms_normal_gpu).
 * and the host twins :275-277 boxes_overlap_bev_cpu / boxes_iou_bev_cpu / boxes_iou3d_cpu (iou3d_cpu.cpp:270-336).
 * mode for pairwise: 0 overlap area (N,5)x(M,5) | 1 BEV IoU (N,5)x(M,5) | 2 3-D IoU (N,7)x(M,7) |
 * 3 3-D IoU with the host twin's convention (no early zero for disjoint z ranges: overlap * 1e-8, iou3d_cpu.cpp:322-333).
 * Boxes are [x1,y1,x2,y2,ry] or [x1,y1,z1,x2,y2,z2,ry]; out is (N,M) row-major float32.
 * NMS: boxes already sorted by descending score; keep (device int64[N]) and num_keep (device
 * int32) are produced ON THE DEVICE (the reference reduces the bitmask on the host). */
int sessd_boxes_pairwise(int mode, const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* out,
                         sessd_stream_t stream);
int sessd_boxes_aligned_overlap_bev(const float* boxes_a, const float* boxes_b, int num, float* out,
                                    sessd_stream_t stream);
/* numba-convention rotated IoU used by the KITTI AP evaluation: det3d/ops/nms/nms_gpu.py:541-577,636-672
 * (rotate_iou_gpu / rotate_iou_gpu_eval). boxes (N,5), query (K,5) [cx,cy,w,l,angle] -> (N,K);
 * criterion -1 IoU | 0 inter/area(query) | 1 inter/area(box) | 2 intersection area. */
int sessd_rotate_iou_eval(const float* boxes, int num_boxes, const float* query, int num_query, int criterion, float* out,
                          sessd_stream_t stream);
/* det3d/datasets/utils/eval.py:324-367 box3d_overlap (rotate_iou_gpu_eval criterion 2 + the numba loop d3_box_overlap_kernel) in
 * one launch: boxes (N,7), query (K,7) rows [loc 3, dims 3, rot], z_axis = height axis among the three (KITTI camera: 1),
 * z_center = position of the location in the height (camera: 1.0); criterion -1 IoU | 0 / box volume | 1 / query volume | 2 raw.
 * float64 rows as the annotations hold them: the rotated BEV part is computed in float32 (as rotate_iou_gpu_eval casts), the
 * height / volume arithmetic in float64 (as the numba loop does). */
int sessd_box3d_overlap_eval(const double* boxes, int num_boxes, const double* query, int num_query, int criterion, int z_axis,
                             double z_center, double* out, sessd_stream_t stream);
/* KITTI AP accumulation (det3d/datasets/kitti/eval.py:121-171,174-319, utils/eval.py:144-278): frames flattened CSR style --
 * overlaps = per frame a (n_det, n_gt) float64 block at ov_off[f]; gt_off / dt_off / dc_off (F+1) row offsets into gt_data (.,5)
 * [bbox 4, alpha], dt_data (.,6) [bbox 4, alpha, score], dc_boxes (.,4), ignored_gt / ignored_det (0 counts | 1 neutral | -1 other).
 * compute_fp == 0: tp_scores (sum n_gt) = score of the detection matched to each ground truth, NaN where none (first pass);
 * compute_fp != 0: stats (F, num_thresholds, 4) = tp, fp, fn, orientation similarity per frame and score threshold.
 * sessd_kitti_thresholds = get_thresholds on scores sorted descending; sessd_kitti_reduce = ordered sum over the frames. */
int sessd_kitti_statistics(const double* overlaps, const long long* ov_off, const int32_t* gt_off, const int32_t* dt_off,
                           const int32_t* dc_off, const double* gt_data, const double* dt_data, const int32_t* ignored_gt,
                           const int32_t* ignored_det, const double* dc_boxes, int num_frames, int metric, double min_overlap,
                           const double* thresholds, int num_thresholds, int compute_fp, int compute_aos, double* tp_scores,
                           double* stats, int32_t* err_flag, sessd_stream_t stream);
int sessd_kitti_reduce(const double* stats, int num_frames, int num_thresholds, double* pr, sessd_stream_t stream);
int sessd_kitti_thresholds(const double* sorted_scores_desc, int num_scores, int num_gt, int num_sample_pts, double* thresholds,
                           int32_t* num_thresholds, sessd_stream_t stream);
/* sessd_nms_sorted modes 3 / 4 = numba rotate_nms_gpu (nms_gpu.py:422-499) / nms_gpu (+1 convention, :36-169) */
size_t sessd_nms_workspace_bytes(int num_boxes);
int sessd_nms_sorted(int mode, const float* boxes, int num_boxes, float thresh, long long* keep, int32_t* num_keep,
                     void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* det3d.ops.nms.nms.non_max_suppression_cpu (det3d/ops/nms/nms_cpu.h:24-70; nms_cpu.py:34-37 nms_cc passes eps = 1) on the
 * device: boxes (N, stride >= 4) [x1,y1,x2,y2,..] in descending-score order, extents widened by eps, suppress at IoU >= thresh. */
int sessd_nms_axis_eps_sorted(const float* boxes, int stride, int num_boxes, float thresh, float eps, long long* keep,
                              int32_t* num_keep, void* workspace, size_t workspace_bytes, sessd_stream_t stream);

/* ------------------------------------------------------------------ sparse 3-D convolution (a4-a8)
 * replace the third-party spconv calls of det3d/models/backbones/scn.py:106-148,179-187:
 * SparseConvTensor / SubMConv3d / SparseConv3d rulebooks (spconv.ops.get_indice_pairs), the
 * convolution itself (spconv indice_conv), BatchNorm1d(eval)+ReLU on .features, and .dense().
 * Sites are (N,4) int32 [b,z,y,x]; live counts stay on the device (n_*_dev), arrays are sized by
 * capacity. A rulebook is output-stationary: nbr[k][o] = input row (or -1), k = (kz*KY+ky)*KX+kx,
 * stored [kernel_volume][n_out_cap]; tile_mask[o/16] has bit k set when any of the 16 sites of the
 * tile has a neighbour through offset k.  The linear key ((b*D + z)*H + y)*W + x is 32 bit: batch * D * H * W must stay below 2^32 - 1
 * (KITTI grid [41,1600,1408]: batch <= 46); the batch size is not an argument, so this is the caller's contract. The fused chain
 * (sessd_sparse_chain_*) checks its own limits. */
int sessd_sparse_hash_build(const int32_t* indices, const int32_t* n_dev, int n_cap, const int32_t* dims3,
                            uint32_t* keys, int32_t* vals, uint32_t capacity, sessd_stream_t stream);
size_t sessd_sparse_downsample_workspace_bytes(int n_in_cap, int kernel_volume, uint32_t out_hash_capacity);
int sessd_sparse_downsample_sites(const int32_t* in_indices, const int32_t* n_in_dev, int n_in_cap,
                                  const int32_t* ksize3, const int32_t* stride3, const int32_t* pad3,
                                  const int32_t* out_dims3, uint32_t* out_keys, int32_t* out_vals,
                                  uint32_t out_capacity, int32_t* out_indices, int n_out_cap, int32_t* n_out_dev,
                                  int32_t* err_flag, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
int sessd_sparse_rulebook(const int32_t* out_indices, const int32_t* n_out_dev, int n_out_cap, const int32_t* ksize3,
                          const int32_t* stride3, const int32_t* pad3, const uint32_t* in_keys, const int32_t* in_vals,
                          uint32_t in_capacity, const int32_t* in_dims3, int32_t* nbr, uint32_t* tile_mask,
                          sessd_stream_t stream);
/* two rulebooks over the SAME output sites in one launch (the strided conv into a level + the submanifold convs on it) */
int sessd_sparse_rulebook_pair(const int32_t* out_indices, const int32_t* n_out_dev, int n_out_cap,
                               const int32_t* ksize3_a, const int32_t* stride3_a, const int32_t* pad3_a,
                               const uint32_t* keys_a, const int32_t* vals_a, uint32_t capacity_a, const int32_t* dims3_a,
                               int32_t* nbr_a, uint32_t* tile_mask_a, const int32_t* ksize3_b, const int32_t* stride3_b,
                               const int32_t* pad3_b, const uint32_t* keys_b, const int32_t* vals_b, uint32_t capacity_b,
                               const int32_t* dims3_b, int32_t* nbr_b, uint32_t* tile_mask_b, sessd_stream_t stream);
/* single-launch variant of sessd_sparse_downsample_sites: rows numbered by an atomic counter (numbering not
 * reproducible run to run; everything computed from it is). Caller pre-clears out_keys/out_vals (0x7F7F7F7F) and
 * *n_out_dev (0). */
int sessd_sparse_downsample_sites_unordered(const int32_t* in_indices, const int32_t* n_in_dev, int n_in_cap,
                                            const int32_t* ksize3, const int32_t* stride3, const int32_t* pad3,
                                            const int32_t* out_dims3, uint32_t* out_keys, int32_t* out_vals,
                                            uint32_t out_capacity, int32_t* out_indices, int n_out_cap,
                                            int32_t* n_out_dev, int32_t* err_flag, sessd_stream_t stream);
/* spconv.SparseConvTensor.dense() (scn.py:184) and its gradient: features (n, channels) at sites indices (n,4) [b,z,y,x]
 * <-> dense (batch, channels, D, H, W) float32, dims3 = (D,H,W) on the host. sessd_sparse_to_dense writes only the site
 * cells: the caller zeroes `dense` first. */
int sessd_sparse_to_dense(const float* features, const int32_t* indices, int n, int channels, const int32_t* dims3,
                          float* dense, sessd_stream_t stream);
int sessd_dense_to_sparse(const float* dense, const int32_t* indices, int n, int channels, const int32_t* dims3,
                          float* features, sessd_stream_t stream);
/* the same two on a table of CAPACITY n_cap rows whose live row count is the device int *n_dev (the capacity-based module path
 * that a captured training graph runs: no host-read counts): rows >= *n_dev are not scattered; their gradient rows are zero */
int sessd_sparse_to_dense_dev(const float* features, const int32_t* indices, int n_cap, const int32_t* n_dev, int channels,
                              const int32_t* dims3, float* dense, sessd_stream_t stream);
int sessd_dense_to_sparse_dev(const float* dense, const int32_t* indices, int n_cap, const int32_t* n_dev, int channels,
                              const int32_t* dims3, float* features, sessd_stream_t stream);

/* ---- the whole strided chain at once (csrc/sparse_sites.hip) ---------------------------------------------------------
 * replaces the per-layer spconv.ops.get_indice_pairs calls behind det3d/models/backbones/scn.py:106-148 (four SparseConv3d,
 * four SubMConv3d groups): the site sets of ALL levels follow from the level-0 sites alone, so three launches (mark the
 * reachable cells of every level in per-level occupancy bit maps; popcount per block; prefix scan + site table) give every
 * level's rows, numbered in ascending (b,z,y,x) order (deterministic; 16 consecutive rows are spatial neighbours), and ONE
 * more launch builds every neighbour table of the chain. `workspace` holds the occupancy maps (rank-annotated after the
 * call; sessd_sparse_chain_rulebooks reads them); clear != 0: the call zeroes it, clear == 0: the caller did (one arena
 * fill per frame). err_flag |= 1 when a level has more active cells than `cap` (rows beyond cap are dropped, as absent). */
size_t sessd_sparse_chain_workspace_bytes(int batch, int n_levels, const sessd_chain_level_t* levels);
int sessd_sparse_chain_sites(const int32_t* indices0, const int32_t* n0_dev, int n0_cap, int batch, int n_levels,
                             const sessd_chain_level_t* levels, void* workspace, size_t workspace_bytes, int clear,
                             int32_t* err_flag, sessd_stream_t stream);
/* level 0 is looked up through its hash (keys0 / vals0 / capacity0, key dims dims0 = (D,H,W) as built by the voxelizer or
 * sessd_sparse_hash_build); deeper levels through the occupancy maps in `workspace`. Same nbr / tile_mask layout as
 * sessd_sparse_rulebook. */
int sessd_sparse_chain_rulebooks(const int32_t* indices0, const int32_t* n0_dev, int n0_cap, const uint32_t* keys0,
                                 const int32_t* vals0, uint32_t capacity0, const int32_t* dims0, int batch, int n_levels,
                                 const sessd_chain_level_t* levels, const void* workspace, int n_jobs,
                                 const sessd_rulebook_job_t* jobs, sessd_stream_t stream);
/* weight (kernel_volume, cin, cout) row-major == spconv's [kz,ky,kx,Cin,Cout] flattened -> MFMA fragment order */
int sessd_sparse_pack_weight(const float* weight, int kernel_volume, int cin, int cout, float* packed,
                             sessd_stream_t stream);
/* Every sparse weight packing of a training iteration (teacher forward, student forward, student data gradient: 41 layers) in ONE
 * launch: jobs_dev = device array of sessd_sparse_pack_job_t with ascending block_start, total_blocks = the launch's grid. */
int sessd_sparse_pack_batch(const sessd_sparse_pack_job_t* jobs_dev, int n_jobs, int total_blocks, sessd_stream_t stream);
/* The packed weight of the (cout -> cin) conv that computes the layer's data gradient, from the same weight tensor:
 * W'[k] = W[k']^T with k' = kernel_volume - 1 - k when reverse_offsets (submanifold layers, whose gradient runs on the forward
 * neighbour table), else k' = k (strided layers, on sessd_sparse_rulebook_transpose's table). cin % 16 == 0, cout % 4 == 0. */
int sessd_sparse_pack_weight_adjoint(const float* weight, int kernel_volume, int cin, int cout, int reverse_offsets,
                                     float* packed, sessd_stream_t stream);
/* out[o] = act((sum_k W[k]^T in[nbr[k][o]]) * scale + shift); scale/shift = folded eval BatchNorm1d (may be NULL).
 * dense_out != NULL: scatter into the pre-zeroed BEV tensor (B, cout*D, H, W), dense_dims3 = (D,H,W).
 * tuning = cout_split + 256 * depth. cout_split: 0 heuristic | 1,2,4 waves per 16-site tile (each computes cout/split
 * channels); depth: 0 default | 2..4 operand register sets (offsets whose rows and weights are in flight, plus the one being
 * multiplied). Results are bit-identical for every tuning. */
int sessd_sparse_conv(const float* in_feat, int cin, const int32_t* nbr, const uint32_t* tile_mask, int kernel_volume,
                      const int32_t* n_out_dev, int n_out_cap, const float* packed_weight, const float* scale,
                      const float* shift, int relu, float* out_feat, int cout, const int32_t* out_indices,
                      float* dense_out, const int32_t* dense_dims3, int tuning, sessd_stream_t stream);
/* The same with OFFSET-PATTERN TILES (perm != NULL): tile_mask = the rulebook job's tile_mask_sorted, perm = its position -> row
 * table (sessd_rulebook_job_t, built by sessd_sparse_chain_rulebooks): inside every group of 256 consecutive rows the sites are
 * grouped into 16-row tiles by their neighbour pattern, so fewer (tile, offset) MFMA steps multiply rows without a neighbour.
 * Rows keep their numbers (no indirection in any lookup); per site the arithmetic is unchanged: results are bit-identical to
 * sessd_sparse_conv on the same tables. */
int sessd_sparse_conv_sorted(const float* in_feat, int cin, const int32_t* nbr, const uint32_t* tile_mask, int kernel_volume,
                             const int32_t* n_out_dev, int n_out_cap, const float* packed_weight, const float* scale,
                             const float* shift, int relu, float* out_feat, int cout, const int32_t* out_indices,
                             float* dense_out, const int32_t* dense_dims3, int tuning, const uint8_t* perm, sessd_stream_t stream);

/* ---- engine-internal site renumbering (no reference counterpart: spconv numbers sites as they come) ---------------
 * EXPERIMENTAL -- compiled, not yet validated on hardware (round 1 ran out of GPU budget); off by default in the engine.
 * Rewrites a level's site table in (batch, z, y) grid-row order so that the 16-site tiles of sessd_sparse_conv hold spatial
 * neighbours. indices (n_cap,4) [b,z,y,x] + feat (n_cap, channels) -> out_indices / out_feat
 * (distinct buffers), and the level's hash (keys ((b*D+z)*H+y)*W+x, dims3 = (D,H,W)) is pointed at the new rows.
 * Convolution results do not depend on the numbering. Order inside a grid row is unspecified. */
size_t sessd_sparse_renumber_workspace_bytes(int batch, const int32_t* dims3);
int sessd_sparse_renumber_sites(const int32_t* indices, const int32_t* n_dev, int n_cap, int batch, const int32_t* dims3,
                                const float* feat, int channels, const uint32_t* hash_keys, int32_t* hash_vals,
                                uint32_t hash_capacity, int32_t* out_indices, float* out_feat, void* workspace,
                                size_t workspace_bytes, sessd_stream_t stream);

/* ---- train-mode BatchNorm1d + ReLU on a sparse level's feature table --------------------------------------
 * EXPERIMENTAL -- compiled, not yet validated on hardware, not wired into the module path.
 * replaces torch.nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU after every sparse conv of det3d/models/backbones/scn.py:103-148
 * in train mode (both networks of the SE-SSD step): batch statistics over the rows < *n_dev, running statistics updated in
 * place (unbiased variance), deterministic reductions (two launches per pass: the statistics block that finishes last adds the
 * partial sums in block order). channels: a power of two <= 256. Rows >= *n_dev of y / dx are written as zeros.
 * WORKSPACE CONTRACT (both layouts): its leading counter words (256 bytes here; 4 bytes per channel rounded up to 256 for the
 * dense layout and sessd_nchw_channel_sum) must be ZERO on entry -- clear a new workspace once -- and are zero again on return. */
size_t sessd_bn_relu_train_workspace_bytes(int channels);
int sessd_bn_relu_train_fwd(const float* x, const int32_t* n_dev, int n_cap, int channels, const float* gamma, const float* beta,
                            float eps, float momentum, int relu, float* running_mean, float* running_var, float* y,
                            float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
int sessd_bn_relu_train_bwd(const float* dy, const float* x, const float* y, const int32_t* n_dev, int n_cap, int channels,
                            const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dx,
                            float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* The same for the dense BEV layout: torch.nn.BatchNorm2d (train mode) + optional ReLU on x (batch, channels, plane = H * W,
 * plane % 4 == 0) -- det3d/models/necks/rpn_v1.py:131-210 in the training step (BatchNorm2d(eps 1e-3, momentum 0.01) + ReLU after
 * each SSFA convolution). Statistics in float64 partial sums with a fixed reduction order. */
size_t sessd_bn2d_relu_train_workspace_bytes(int channels);
int sessd_bn2d_relu_train_fwd(const float* x, int batch, int channels, int plane, const float* gamma, const float* beta, float eps,
                              float momentum, int relu, float* running_mean, float* running_var, float* y, float* save_mean,
                              float* save_invstd, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
int sessd_bn2d_relu_train_bwd(const float* dy, const float* x, const float* y, int batch, int channels, int plane,
                              const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dx,
                              float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* The same gradients without the forward output y: the ReLU mask is re-derived from x with the forward's gamma / beta (the
 * forward computes its pre-ReLU value with the same roundings, so the mask is the same bit for bit); both launches read
 * one tensor less. */
int sessd_bn2d_relu_train_bwd_x(const float* dy, const float* x, int batch, int channels, int plane, const float* gamma,
                                const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dx,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* ---- SyncBN form of the four passes above, for world size > 1: the reference's distributed path converts every BatchNorm to SyncBN
 * (det3d/torchie/apis/train_sessd.py:286-294, apex convert_syncbn_model; in-tree twin det3d/ops/syncbn/syncbn.py:37-103) -- batch
 * statistics over the batches of ALL ranks. Each pass is split where the ranks must talk: *_stats writes THIS rank's float64 totals
 * (forward: [sum x (C), sum x^2 (C), N]; backward: [sum dz (C), sum dz * xhat (C)], and the LOCAL dgamma / dbeta, which are averaged
 * with all other parameter gradients), the caller all-reduces (sums) them over RCCL, *_apply finalises (mean / invstd / running
 * statistics with the global N; backward: dx with the global sums and 1 / N_global) and runs the apply launch. fwd_sums of
 * *_bwd_apply = the forward's all-reduced totals. Same workspace contract as above; scratch: sessd_bn_sync_scratch_bytes(channels).
 * With one rank the result equals the fused form bit for bit. */
size_t sessd_bn_sync_scratch_bytes(int channels);
int sessd_bn_relu_train_stats(const float* x, const int32_t* n_dev, int n_cap, int channels, double* sums, void* workspace,
                              size_t workspace_bytes, sessd_stream_t stream);
int sessd_bn_relu_train_apply(const float* x, const int32_t* n_dev, int n_cap, int channels, const float* gamma, const float* beta,
                              float eps, float momentum, int relu, const double* sums, float* running_mean, float* running_var,
                              float* y, float* save_mean, float* save_invstd, sessd_stream_t stream);
int sessd_bn_relu_train_bwd_stats(const float* dy, const float* x, const float* y, const int32_t* n_dev, int n_cap, int channels,
                                  const float* save_mean, const float* save_invstd, int relu, float* dgamma, float* dbeta,
                                  double* sums, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
int sessd_bn_relu_train_bwd_apply(const float* dy, const float* x, const float* y, const int32_t* n_dev, int n_cap, int channels,
                                  const float* gamma, const float* save_mean, const float* save_invstd, int relu, const double* sums,
                                  const double* fwd_sums, float* dx, void* scratch, size_t scratch_bytes, sessd_stream_t stream);
int sessd_bn2d_relu_train_stats(const float* x, int batch, int channels, int plane, double* sums, void* workspace,
                                size_t workspace_bytes, sessd_stream_t stream);
int sessd_bn2d_relu_train_apply(const float* x, int batch, int channels, int plane, const float* gamma, const float* beta, float eps,
                                float momentum, int relu, const double* sums, float* running_mean, float* running_var, float* y,
                                float* save_mean, float* save_invstd, sessd_stream_t stream);
int sessd_bn2d_relu_train_bwd_stats(const float* dy, const float* x, const float* y, int batch, int channels, int plane,
                                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                    float* dgamma, float* dbeta, double* sums, void* workspace, size_t workspace_bytes,
                                    sessd_stream_t stream);
int sessd_bn2d_relu_train_bwd_apply(const float* dy, const float* x, const float* y, int batch, int channels, int plane,
                                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                    const double* sums, const double* fwd_sums, float* dx, void* scratch, size_t scratch_bytes,
                                    sessd_stream_t stream);

/* ---- training data path, point-level work -----------------------------------------------------------
 * replaces det3d/core/bbox/geometry.py:215-276 points_in_convex_polygon_3d_jit (numba) as used by
 * box_np_ops.points_in_rbbox :1152, sampler/preprocess.py:645 (per-object noise) and sa_da_v2.py:65-74 (pyramids).
 * points (num_points, point_stride) float32; planes (num_bodies, faces, 4) float32 [nx,ny,nz,d], inward normals, from the host
 * (geometry.py:352-377 arithmetic); out_mask (num_points, ceil(num_bodies/32)) uint32, bit set = strictly inside
 * (x*nx + y*ny + z*nz + d < 0 for every face, float32, left to right, no contraction: identical to the numba loop). */
/* HOST functions (no device work): the box-level decisions of the same stage -- det3d/core/sampler/preprocess.py:944-1027
 * box_collision_test and :579-611 noise_per_box (numba in the reference; sequential over tens of boxes, so they stay on the
 * host even when the cloud is on the device). boxes / qboxes / corners: (n, 4, 2) BEV corners, float32 (is_f32) or float64;
 * the arithmetic is that of the numpy mirror in that precision, decisions identical. out (n, k) uint8. noise_per_box: corners
 * are updated in place with the accepted moves; loc_xy (n, tries, 2), sin_r / cos_r (n, tries) float64; chosen (n) int64 = first
 * collision-free candidate or -1. */
int sessd_box_collision_host(const void* boxes, int n, const void* qboxes, int k, int is_f32, int clockwise, uint8_t* out);
int sessd_noise_per_box_host(void* corners, const void* centers, const uint8_t* valid, const double* loc_xy, const double* sin_r,
                             const double* cos_r, int n, int tries, int is_f32, int64_t* chosen);
int sessd_points_in_bodies(const float* points, int num_points, int point_stride, const float* planes, int num_bodies,
                           int faces, uint32_t* out_mask, sessd_stream_t stream);
/* det3d/core/sampler/preprocess.py:544-560 points_transform_ (the point side of noise_per_object_v4_): in place, every point takes
 * the rigid motion of the FIRST valid box that contains it (membership fused: same planes and comparison as above) -- rotation
 * about the box centre, then the translation, rounded step by step like the reference's in-place float32 row updates with
 * float64 operands. planes (num_boxes, 6, 4), centers / loc (num_boxes, 3) float64, sincos (num_boxes, 2) [sin, cos], valid
 * (num_boxes) bytes; num_boxes <= 128. */
int sessd_points_rigid_moves(float* points, int num_points, int point_stride, const float* planes, const double* centers,
                             const double* loc, const float* sincos, const uint8_t* valid, int num_boxes, sessd_stream_t stream);
/* preprocess.py:896-945 random_flip_v2 + global_rotation_v3 + global_scaling_v3 applied to the points in one pass (the draws
 * themselves stay on the host); raw_copy != NULL: the cloud as it was before (points_raw, pipelines/preprocess.py:130-134). */
int sessd_points_global_transform(float* points, int num_points, int point_stride, int flip, float sin_angle, float cos_angle,
                                  float scale, float* raw_copy, sessd_stream_t stream);
/* order-preserving compaction by a keep byte per point (GT-AUG removal of covered points pipelines/preprocess.py:102-105,
 * shape-aware dropout): out rows + *n_out on the device, ready for sessd_voxelize_frame. */
size_t sessd_points_compact_workspace_bytes(int num_points);
int sessd_points_compact(const float* points, const uint8_t* keep, int num_points, int point_stride, float* out,
                         int out_capacity, int32_t* n_out, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* det3d/datasets/utils/sa_da_v2.py:76-205 pyramid_augment_v0, thinning step (`ifp_sample` of jackd/ifp-sample over all-pairs
 * neighbourhoods = plain iterative farthest-point sampling): k indices of the n points, starting at point 0, the point farthest
 * from the selection next (float64 Euclidean distance of the float32 coordinates, lowest index on ties); k <= n <= 4096. */
int sessd_farthest_point_sample(const float* points, int num_points, int point_stride, int k, int32_t* out_indices,
                                sessd_stream_t stream);

/* ---- sparse conv backward (spconv's indice_conv backward as differentiated by the SE-SSD training
 * step, det3d/torchie/trainer/trainer_sessd.py:250-275 through det3d/models/backbones/scn.py:106-148) -------------
 * Data gradient: dx = sessd_sparse_conv(dy, nbr_t, tile_mask_t, weights W_k^T) over the INPUT sites, with
 * nbr_t[k][i] = j <=> nbr[k][j] = i built here (collision-free scatter; nbr_t (kv, n_in_cap), tile_mask_t
 * (ceil(n_in_cap/16)) are cleared by the call). */
int sessd_sparse_rulebook_transpose(const int32_t* nbr, int kernel_volume, const int32_t* n_out_dev, int n_out_cap,
                                    int n_in_cap, int32_t* nbr_t, uint32_t* tile_mask_t, sessd_stream_t stream);
size_t sessd_sparse_conv_wgrad_workspace_bytes(int kernel_volume, int cin, int cout);
/* Weight gradient: grad_weight (kv, cin, cout) = sum over rulebook pairs of in_feat[nbr[k][j]] (outer) grad_out[j];
 * deterministic (site chunks summed in order on the matrix cores, then <= 64 partials in order). */
int sessd_sparse_conv_wgrad(const float* in_feat, int cin, const float* grad_out, int cout, const int32_t* nbr,
                            const uint32_t* tile_mask, int kernel_volume, const int32_t* n_out_dev, int n_out_cap,
                            float* grad_weight, void* workspace, size_t workspace_bytes, sessd_stream_t stream);

/* ---- parameter update of the training step: replaces the per-tensor host loops of
 * det3d/torchie/trainer/hooks/optimizer.py:50-53 (clip_grad_norm_), det3d/solver/fastai_optim.py:155-176
 * (OptimWrapper.step with true_wd, then torch.optim.Adam.step) and det3d/torchie/trainer/trainer_sessd.py:315-318
 * (EMA teacher) by two launches over FLAT float32 buffers. All pointers device, 16-byte aligned. */
size_t sessd_grad_clip_workspace_bytes(void);
/* out2 (device float[2]) = { ||grad||_2, min(1, max_norm/(norm + 1e-6)) }; max_norm <= 0: coefficient 1. No host sync. */
int sessd_grad_clip_coef(const float* grad, size_t n, float max_norm, void* workspace, size_t workspace_bytes, float* out2,
                         sessd_stream_t stream);
/* g' = grad*clip2[1] (clip2 may be NULL); p *= 1 - weight_decay*lr; Adam(beta1, beta2, eps) step number `step` >= 1 on
 * (param, exp_avg, exp_avg_sq); ema_param = ema_alpha*ema_param + (1-ema_alpha)*param (ema_param may be NULL). */
int sessd_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema_param, size_t n,
                        double lr, double weight_decay, double beta1, double beta2, double eps, int step,
                        const float* clip2, double ema_alpha, sessd_stream_t stream);
/* The schedule and the update with DEVICE-resident constants, so that a whole training iteration can be one captured graph
 * (a replay must use the current learning rate, not the captured one): sessd_one_cycle_args reads and advances the device
 * iteration counter *global_step and writes the nine Adam / EMA constants of that iteration (OneCycle of
 * det3d/solver/learning_schedules_fastai.py:70-95 with config.py:260's parameters; optimizer step t = *global_step + 1; EMA
 * alpha of trainer_sessd.py:316) to args9 and (lr, momentum) to lr_mom2 (may be NULL); sessd_adam_ema_step_dev consumes args9. */
int sessd_one_cycle_args(int32_t* global_step, int total_steps, double lr_max, double mom_hi, double mom_lo, double div_factor,
                         double pct_start, double weight_decay, double beta2, double eps, float* args9, float* lr_mom2,
                         sessd_stream_t stream);
int sessd_adam_ema_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema_param, size_t n,
                            const float* args9, const float* clip2, sessd_stream_t stream);
/* out[0] = scale * (sum of the n floats at x): deterministic two-stage reduction (double accumulation), workspace of
 * sessd_grad_clip_workspace_bytes(). For losses inside a CAPTURED iteration: torch's multi-block reductions (sum / mean of a large
 * tensor) clear a semaphore buffer with a memset, and a memset NODE of a replayed hipGraph is broken on ROCm 7.2 / gfx950 (correct
 * on the first replay, garbage afterwards), so such a reduction returns stale or foreign values from the second replay on. */
int sessd_sum_f32(const float* x, size_t n, float scale, void* workspace, size_t workspace_bytes, float* out,
                  sessd_stream_t stream);
/* out[c] = sum over images and pixels of x (B, C, plane): a conv's bias gradient, on the statistics pass of the train-mode
 * BatchNorm2d kernels (plane % 4 == 0; workspace of sessd_bn2d_relu_train_workspace_bytes(channels)); same reason as above */
int sessd_nchw_channel_sum(const float* x, int batch, int channels, int plane, float* out, void* workspace,
                           size_t workspace_bytes, sessd_stream_t stream);

/* ---- dense conv backward (training step): data gradients are launches of the forward entry points
 * above with re-packed weights (3x3 s1 <-> flipped 3x3 s1, 3x3 s2 <-> sessd_deconv2d_s2_mfma, 1x1 <-> 1x1); the weight
 * gradient of Conv2d(cin, cout, k, stride, padding k/2) is this pixel-reduction GEMM (deterministic: <= 64 row chunks
 * summed in order). k in {1,3}, stride in {1,2}, wout % 8 == 0. For ConvTranspose2d(3, s2, p1, op1) swap the roles
 * (input := its grad_out, grad_out := its input): the result is its (Cin, Cout, 3, 3) weight gradient.
 * Replaces the ATen/MIOpen backward of det3d/models/necks/rpn_v1.py:135-235 under trainer_sessd.py:250-275. */
/* Every dense weight packing of a training iteration (sessd_conv2d_pack_taps / sessd_conv3x3_winograd_pack jobs) in ONE launch. */
int sessd_dense_pack_batch(const sessd_dense_pack_job_t* jobs_dev, int n_jobs, int total_blocks, sessd_stream_t stream);
size_t sessd_conv2d_wgrad_workspace_bytes(int cout, int cin, int ksize);
int sessd_conv2d_wgrad(const float* input, int batch, int cin, int hin, int win, const float* grad_out, int cout, int hout,
                       int wout, int ksize, int stride, float* grad_weight, void* workspace, size_t workspace_bytes,
                       sessd_stream_t stream);

/* The same weight gradient for Conv2d(cin, cout, 3, stride 1, padding 1) computed in the Winograd F(2x2,3x3) domain
 * (dU_xi = sum over 2x2 output tiles of (A dY A^T)_xi (B^T d B)_xi on the f32 matrix cores, then G^T dU G: 16 instead of 36
 * products per tile): cin, cout multiples of 64, h, w even, w >= 16. _workspace_bytes returns 0 for a shape it does not cover
 * (use sessd_conv2d_wgrad). Deterministic (split over tile chunks, summed in chunk order); equal to sessd_conv2d_wgrad up to
 * Winograd rounding (~1e-6 of the result's scale). */
size_t sessd_conv3x3_wgrad_winograd_workspace_bytes(int batch, int cin, int cout, int h, int w);
int sessd_conv3x3_wgrad_winograd(const float* input, int batch, int cin, int h, int w, const float* grad_out, int cout,
                                 float* grad_weight, void* workspace, size_t workspace_bytes, sessd_stream_t stream);

/* ---- ODIoU loss as one differentiable device op: det3d/models/losses/odious.py:837-900 (odiou_3D and
 * the host-side numpy / scipy functions it composes, :15-643). gboxes (targets), qboxes (predictions) (n,7) float32
 * [x,y,z,w,l,h,r] -> term (n,) = 1 - IoU3D + centre-distance / enclosing-diagonal + 1.25(1 - |cos dr|) and grad_q (n,7) =
 * d term / d qboxes (float64 forward-mode differentiation inside the kernel). The reference's loss is
 * 2 * sum(weights * term) / batch_size. */
int sessd_odiou3d(const float* gboxes, const float* qboxes, int n, float* term, float* grad_q, sessd_stream_t stream);

/* ---- The SE-SSD training loss in capacity form (BASELINE configs[2]): values AND the gradient with respect to
 * the student's four head outputs in six launches, every count on the device -- replaces, for the single-task car head,
 * det3d/models/bbox_heads/mg_head_sessd.py:706-808 (MultiGroupHead.loss), :810-890 (get_model_ema_loss), :618-704
 * (consistency_loss), :573-607 (nn_distance '10'), :525-571 (prepare_loss_weights, NormByNumPositives), the loss classes of
 * det3d/models/losses/losses.py:146-203,364-418,489-531, odiou_3D (losses/odious.py:837-900), boxes_aligned_iou3d_gpu /
 * boxes_iou_bev_gpu (core/iou3d/iou3d_utils.py:32-52,197-252) and trainer_sessd.py:267 (loss += consistency * weight).
 * student / teacher: head outputs + targets (the teacher's are labels_raw / reg_targets_raw / anchors_raw; log terms only).
 * anchors0 (A,7): example["anchors"][0][0], with which the consistency loss decodes every sample (mg_head_sessd.py:649-650).
 * transformation (B,5) device float32 [flipped, cos(noise_rotation), sin(noise_rotation), noise_rotation, noise_scale];
 * consistency_weight: one device float. grad_* (device, same shapes as the student's head outputs) are WRITTEN:
 * d(loss + consistency_weight * consistency) / d(head output). record: 64 device floats --
 *   [0] total loss, [1] loss without consistency, [2] cls_loss_reduced, [3] loc_loss_reduced (logged only), [4] dir loss,
 *   [5] iou_pred_loss, [6] ious_loss (ODIoU), [7] consistency_loss (unweighted), [8] cls_pos_loss, [9] cls_neg_loss,
 *   [10..16] loc_loss_elem, [17] num_pos, [18] num_neg (sample 0), [19..21] box / score / IoU consistency parts, [22] matched boxes;
 *   [24] loss_ema, [26..42] the teacher's [2..18]; [48] overflow flags (1: positives, 2: candidates), [49] / [50] positives of
 *   the batch (student / teacher), [51] / [52] most consistency candidates in one sample (student / teacher).
 * Deterministic (ordered sums, no float atomics); no host synchronisation: capturable in a hipGraph. */
size_t sessd_head_loss_workspace_bytes(const sessd_head_loss_cfg_t* cfg);
int sessd_head_loss(const sessd_head_loss_cfg_t* cfg, const sessd_head_loss_net_t* student, const sessd_head_loss_net_t* teacher,
                    const float* anchors0, const float* transformation, const float* consistency_weight, float* grad_box,
                    float* grad_cls, float* grad_dir, float* grad_iou, float* record, void* workspace, size_t workspace_bytes,
                    sessd_stream_t stream);

/* ---- anchor target assignment: det3d/datasets/pipelines/preprocess.py:236-358 (AssignTarget) ->
 * det3d/core/anchor/target_assigner.py:68-136 -> det3d/core/anchor/target_ops_v3.py:11-137 (create_target_np) with the
 * nearest-IoU similarity (region_similarity.py:85-98) and second_box_encode (box_np_ops.py:52-110). anchors (n,7), gt_boxes
 * (m,7) [x,y,z,w,l,h,r], gt_classes (m,) or NULL (all 1), m <= 128. labels in {-1 ignore, 0 background, class};
 * gt_id = assigned ground-truth index of the foreground anchors, -1 elsewhere. */
size_t sessd_assign_targets_workspace_bytes(int num_anchors);
int sessd_assign_targets(const float* anchors, int num_anchors, const float* gt_boxes, const int32_t* gt_classes, int num_gt,
                         float matched_threshold, float unmatched_threshold, int32_t* labels, float* bbox_targets,
                         float* bbox_outside_weights, int32_t* gt_id, void* workspace, size_t workspace_bytes,
                         sessd_stream_t stream);

/* ------------------------------------------------------------------ dense BEV neck + heads (a9-a10)
 * replace the ATen/cuDNN conv2d, conv_transpose2d, batch_norm, relu, softmax calls made by
 * det3d/models/necks/rpn_v1.py:220-235 (SSFA.forward; RPN.forward :107-116 uses the same layers) and
 * det3d/models/bbox_heads/mg_head_sessd.py:217-230 (Head.forward, four 1x1 convs).
 * Activations are NCHW float32. One launcher covers 3x3 s1, 3x3 s2, 1x1 and the four output-parity
 * classes of ConvTranspose2d(3, stride 2, padding 1, output_padding 1):
 *   input pixel = (y*in_mul + taps_dy[t], x*in_mul + taps_dx[t]), (y,x) in [0,tile_h) x [0,tile_w)
 *   output pixel = (y*out_mul + out_py, x*out_mul + out_px)
 *   out = act(conv * scale[c] + shift[c]) (+ residual)
 * wpk: packed weights [cin/2][ntaps][2][cout_pad], cout_pad = roundup(cout,32), element
 *   wpk[kp][t][h][co] = W[co][2*kp+h][tap t]. tile_cfg 0..5 selects the wave/workgroup tiling. */
int sessd_conv2d_mfma(const float* in, int batch, int cin, int hin, int win, const float* wpk, int ntaps,
                      const int* taps_dy, const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out, int cout,
                      int hout, int wout, int out_mul, int out_py, int out_px, const float* scale, const float* shift,
                      int relu, const float* residual, int tile_cfg, sessd_stream_t stream);
/* Conv2d(cin, cout, 3, stride 1, padding 1) by fused Winograd F(2x2,3x3) on the f32 matrix cores: 2.25x fewer MFMAs
 * than sessd_conv2d_mfma for the same layer (not bit-identical to it: Winograd rounding, ~1e-6 of the output scale).
 * upk = U = G g G^T packed [cin/2][4][2][cout_pad][4] = [k-step][xi / 4][channel parity][cout][xi % 4], xi = 4*row + col.
 * Even H and W, cin % 8 == 0. variant 0 / 1: operands fetched one / two rounds (8 / 16 input channels) ahead of their
 * use (variant 1: cin % 32 == 0) -- same arithmetic, same results. */
int sessd_conv3x3_winograd(const float* in, int batch, int cin, int h, int w, const float* upk, float* out, int cout,
                           const float* scale, const float* shift, int relu, const float* residual, int variant,
                           sessd_stream_t stream);
/* Weight packing on the device, one launch each (ops.pack_conv2d / pack_deconv2d_s2 / pack_winograd*; the training step re-packs every
 * dense conv weight for the teacher forward, the student forward and the student data gradient of every iteration).
 * sessd_conv2d_pack_taps: out [cin/2][ntaps][2][cout_pad32], out[kp][t][h][o] = w[o * out_stride + (2 kp + h) * in_stride +
 * tap_offsets[t]] (element strides / offsets, ntaps <= 16, tap_offsets a HOST array): a transposed, flipped or tap-selected view
 * of the stored weight without an intermediate tensor. sessd_conv3x3_winograd_pack: U = G g G^T of the 3x3 filters of the same
 * kind of view (flip != 0: the adjoint layer), written in the layout of sessd_conv3x3_winograd (layout 0) or of
 * sessd_conv3x3_winograd_sk shape 0 / 1 (layout 1 / 2), padding included. */
int sessd_conv2d_pack_taps(const float* w, long long out_stride, long long in_stride, const int* tap_offsets, int ntaps, int cout,
                           int cin, float* out, sessd_stream_t stream);
int sessd_conv3x3_winograd_pack(const float* w, long long out_stride, long long in_stride, int flip, int cout, int cin, int layout,
                                float* out, sessd_stream_t stream);
/* Second generation of the above (csrc/dense_wino_sk.hip): a workgroup handles 32 tiles x ALL couts of a unit (the patch
 * transform runs once per tile block instead of once per 32-cout block) and the rounds of all units are dealt out "stream-K" in
 * equal shares to `workgroups` persistent workgroups (a multiple of 8; 0 = the shape's default), units cut by a share boundary
 * being finished by whichever part arrives last. shape 0: 8 waves x 128 couts per workgroup, one per CU; shape 1: 4 waves x 64
 * couts, two per CU. upk = U = G g G^T packed [ceil(cout / C)][cin/2][NW][2][32][C/32][16/NW], (NW, C) = (8, 128) / (4, 64);
 * cin % (2 NW) == 0, even H and W. workspace: sessd_conv3x3_winograd_sk_workspace_bytes(...) bytes, zeroed ONCE by the caller
 * (the kernel leaves its counters zero), not shared between launches that may run concurrently. */
size_t sessd_conv3x3_winograd_sk_workspace_bytes(int batch, int h, int w, int cout, int shape, int workgroups);
int sessd_conv3x3_winograd_sk(const float* in, int batch, int cin, int h, int w, const float* upk, float* out, int cout,
                              const float* scale, const float* shift, int relu, const float* residual, void* workspace,
                              size_t workspace_bytes, int shape, int workgroups, sessd_stream_t stream);
/* Several layers of ONE shape in one launch (conv_0 / conv_1 of the SSFA neck, rpn_v1.py:201-210): the batch dimension is nsets
 * consecutive groups of batch / nsets elements, group s convolved with weight set s (upk = nsets packings back to back, scale /
 * shift = nsets x cout): one round list, one pipeline fill and one tail instead of nsets. nsets == 1 is the call above. */
int sessd_conv3x3_winograd_sk_sets(const float* in, int batch, int nsets, int cin, int h, int w, const float* upk, float* out,
                                   int cout, const float* scale, const float* shift, int relu, const float* residual,
                                   void* workspace, size_t workspace_bytes, int shape, int workgroups, sessd_stream_t stream);
/* ACTIVE-TILE mode of the stream-K kernel (csrc/dense_wino_sk.hip, LIST = true; csrc/dense_active.hip). The BEV map entering the
 * neck (`.dense()` of the last sparse level, det3d/models/backbones/scn.py:179-183) is zero outside the sparse backbone's sites, so
 * the first three layers of bottom_up_block_0 (rpn_v1.py:135-148) compute the same per-channel constant in every 2x2-output tile
 * whose 4x4 input patch holds no non-constant pixel (82 % / 71 % / 61 % of the tiles on a 20 k-point scan):
 *   sessd_bev_tile_activity     tile masks + ordered tile lists (entry image * tiles + tile) + device counts of a chain of 3x3 layers
 *                               given as HOST steps (0 = stride-1 layer taking the next slot, 1 = stride-2 layer computed
 *                               everywhere, 2 = stride-2 layer that takes a slot itself -- the 2x2 tiles of ITS output with a
 *                               non-constant pixel, 3 = stride-2 transposed conv on the current map with a residual of the
 *                               resolution before the halving, a slot of 2x2 tiles of its INPUT, 4 = no layer: the map
 *                               becomes the OUTPUT of the step-3 transposed conv in front of it -- twice the resolution,
 *                               non-constant in the 4x4 blocks of that step's tiles, constant PER OUTPUT PARITY CLASS elsewhere --
 *                               so that a following 0 is a 3x3 layer behind the transposed convs: rpn_v1.py:135-160 + 224 is
 *                               {0, 0, 0, 2, 0, 0, 3}, with conv_0 / conv_1 (:200-210) {.., 3, 4, 0}; <= 8 slots, <= 10 steps),
 *                               from the (image, z, y, x) rows of the last sparse level
 *   sessd_fill_inactive_tiles   out[b][co][tile] = value[co] (the layer's constant, computed by the host from the folded weights;
 *                               one value per output parity class for job.tile = 4 / 6) in the tiles nobody computes, up to 12
 *                               layers per launch
 *   sessd_conv3x3_winograd_sk_active   sessd_conv3x3_winograd_sk over the listed tiles only (same packed U, same workspace; the
 *                               shares of the round list are sized on the device, workgroups beyond rounds / min_rounds exit;
 *                               min_rounds = -k: whole-unit shares of at least k units, no unit cut, no partial sums in memory)
 * Results equal the dense layer's to float32 rounding (tests/test_dense_active_gpu.py); replaces nothing in the reference -- it is
 * how this path avoids arithmetic on constants that ATen's dense conv performs. */
size_t sessd_bev_tile_activity_workspace_bytes(int batch, int n_layers);
int sessd_bev_tile_activity(const int32_t* indices, const int32_t* n_dev, int n_cap, int batch, int h, int w, const int32_t* steps,
                            int n_steps, uint64_t* tile_mask, int32_t* tile_list, int32_t* n_list, int list_cap, void* workspace,
                            size_t workspace_bytes, sessd_stream_t stream);
int sessd_fill_inactive_tiles(const sessd_fill_tiles_job_t* jobs, int n_jobs, int batch, sessd_stream_t stream);
int sessd_conv3x3_winograd_sk_active(const float* in, int batch, int cin, int h, int w, const float* upk, float* out, int cout,
                                     const float* scale, const float* shift, int relu, const float* residual,
                                     const int32_t* tile_list, const int32_t* n_list, int list_cap, int min_rounds, void* workspace,
                                     size_t workspace_bytes, int shape, int workgroups, sessd_stream_t stream);
/* The convolutions that are NOT 3x3 stride 1 (stride-2 3x3, 1x1, the four output-parity classes of the stride-2 transposed conv;
 * rpn_v1.py:150-210) as an LDS-tiled implicit GEMM, stream-K over `workgroups` persistent workgroups (a multiple of 8, 0 = one
 * per CU): csrc/dense_conv_sk.hip, tile_cfg 30 of ops.conv2d. nclass (1..4) convolutions that share input, shapes and epilogue
 * but not weights / taps / output phase run as ONE launch. Class c: wpk[c] = device pointer from sessd_conv2d_sk_pack
 * ([ceil(cout/128)][cin/16][ntaps][2048]: the kernel's LDS image, element ((h*2+q)*128 + i)*4 + e of a chunk =
 * w[(cg*128 + i) * out_stride + (cb*16 + 8h + 4q + e) * in_stride + tap_offsets[t]]), ntaps[c] <= 9 taps with input offsets
 * taps_dy / taps_dx[9 c + t] (HOST ints); tile-space pixel (y, x) of tile_h x tile_w reads input (y*in_mul + dy, x*in_mul + dx)
 * and writes output (y*out_mul + out_py[c], x*out_mul + out_px[c]). cin % 16 == 0. Exact float32 arithmetic; the summation
 * order (hence the last bits) is fixed by (shapes, batch, workgroups), not by timing. workspace:
 * sessd_conv2d_sk_workspace_bytes(...) bytes, zeroed ONCE by the caller, not shared between concurrently running launches. */
size_t sessd_conv2d_sk_workspace_bytes(int batch, int tile_h, int tile_w, int cout, int nclass, int workgroups);
int sessd_conv2d_sk_pack(const float* w, long long out_stride, long long in_stride, const int* tap_offsets, int ntaps, int cout,
                         int cin, float* out, sessd_stream_t stream);
int sessd_conv2d_sk(const float* in, int batch, int cin, int hin, int win, int nclass, const float* const* wpk,
                    const int* ntaps, const int* taps_dy, const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out,
                    int cout, int hout, int wout, int out_mul, const int* out_py, const int* out_px, const float* scale,
                    const float* shift, int relu, const float* residual, void* workspace, size_t workspace_bytes, int workgroups,
                    sessd_stream_t stream);
/* ACTIVE-TILE mode of that launch (the stride-2 conv that opens bottom_up_block_1 and the 1x1 trans_0 / trans_1 of the SSFA neck,
 * rpn_v1.py:150-152,163-172, whose inputs are a per-channel constant away from the sparse sites): only the 2x2 tiles of the TILE
 * SPACE listed in tile_list[0 .. min(*n_list, list_cap)) (entries image * (tile_h/2 * tile_w/2) + tile, count on the device:
 * sessd_bev_tile_activity; a 1x1 layer takes the list of the layer that produced its input) are computed, the other output
 * pixels are left alone (sessd_fill_inactive_tiles). Even tile_h, tile_w; the whole batch tensor is addressed through 32-bit
 * offsets. Shares of the round list are at least min_rounds rounds; same packed weights and workspace as sessd_conv2d_sk. */
int sessd_conv2d_sk_active(const float* in, int batch, int cin, int hin, int win, int nclass, const float* const* wpk,
                           const int* ntaps, const int* taps_dy, const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out,
                           int cout, int hout, int wout, int out_mul, const int* out_py, const int* out_px, const float* scale,
                           const float* shift, int relu, const float* residual, const int32_t* tile_list, const int32_t* n_list,
                           int list_cap, int min_rounds, void* workspace, size_t workspace_bytes, int workgroups,
                           sessd_stream_t stream);
/* ConvTranspose2d(cin, cout, 3, stride 2, padding 1, output_padding 1) as ONE launch over its four output-parity
 * classes (py,px) = (0,0),(0,1),(1,0),(1,1) with 1,2,2,4 taps: wpk4[c] packed like above, taps_dy4/taps_dx4 are
 * 4 rows of 4 ints. input (B,cin,hin,win) -> output (B,cout,2*hin,2*win); cin % 8 == 0. */
int sessd_deconv2d_s2_mfma(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4,
                           const int* ntaps4, const int* taps_dy4, const int* taps_dx4, float* out, int cout,
                           const float* scale, const float* shift, int relu, const float* residual, int tile_cfg,
                           sessd_stream_t stream);
/* Two such layers applied to the SAME input (deconv_block_0 / deconv_block_1 of the SSFA neck, rpn_v1.py:224-226) as one launch
 * over their 2 x 4 parity classes; per layer wpk4 / out / scale / shift / residual as above, shared tap tables. tile_cfg 3, 4, 11
 * or 12. The same bits as two single-layer launches. */
int sessd_deconv2d_s2_mfma_pair(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4_a,
                                const float* const* wpk4_b, const int* ntaps4, const int* taps_dy4, const int* taps_dx4,
                                float* out_a, float* out_b, int cout, const float* scale_a, const float* shift_a,
                                const float* scale_b, const float* shift_b, int relu, const float* residual_a,
                                const float* residual_b, int tile_cfg, sessd_stream_t stream);
/* ACTIVE-TILE mode of sessd_conv2d_mfma (1x1 layers: trans_0 / trans_1, rpn_v1.py:163-172) and of sessd_deconv2d_s2_mfma_pair
 * (rpn_v1.py:175-199, 224-226) on maps that are a per-channel constant away from the sparse sites (csrc/dense_active.hip): only the
 * 2x2 tiles of the TILE SPACE listed in tile_list[0 .. min(*n_list, list_cap)) (entries image * (tile_h/2 * tile_w/2) + tile, count
 * on the device: sessd_bev_tile_activity) are computed -- for the transposed convs the tile space is the INPUT map (step 3 of the
 * activity program), a listed tile gives a 4x4 block of both outputs --, the other output pixels are left alone
 * (sessd_fill_inactive_tiles, tile = 4 with one constant per output parity class for the transposed convs). The launch is sized
 * for the whole map, workgroups beyond the device count leave at once. Four effective taps (a 1x1 layer with cin % 8 == 0 or a
 * class of the transposed conv), tile_cfg 3 / 4 / 11 / 12, even tile_h / tile_w, the whole batch inside 32-bit buffer offsets.
 * Per computed pixel the code of the plain launch: the same bits. */
int sessd_conv2d_mfma_active(const float* in, int batch, int cin, int hin, int win, const float* wpk, int ntaps, const int* taps_dy,
                             const int* taps_dx, int in_mul, int tile_h, int tile_w, float* out, int cout, int hout, int wout,
                             int out_mul, int out_py, int out_px, const float* scale, const float* shift, int relu,
                             const float* residual, const int32_t* tile_list, const int32_t* n_list, int list_cap, int tile_cfg,
                             sessd_stream_t stream);
int sessd_deconv2d_s2_mfma_pair_active(const float* in, int batch, int cin, int hin, int win, const float* const* wpk4_a,
                                       const float* const* wpk4_b, const int* ntaps4, const int* taps_dy4, const int* taps_dx4,
                                       float* out_a, float* out_b, int cout, const float* scale_a, const float* shift_a,
                                       const float* scale_b, const float* shift_b, int relu, const float* residual_a,
                                       const float* residual_b, const int32_t* tile_list, const int32_t* n_list, int list_cap,
                                       int tile_cfg, sessd_stream_t stream);
/* rpn_v1.py:227-233: softmax over the two 1-channel weight maps and the weighted sum of x0, x1 */
int sessd_ssfa_fuse(const float* x0, const float* x1, const float* w0, const float* w1, float bn_scale0,
                    float bn_shift0, float bn_scale1, float bn_shift1, int batch, int channels, int num_pixels,
                    float* out, sessd_stream_t stream);
/* The head outputs' layout change (mg_head_sessd.py:217-230: `.permute(0, 2, 3, 1).contiguous()` per 1x1 conv) for the fused
 * head tensor: planar (batch, channels, plane) -> parts[k] (batch, plane, sizes[k]), k < n_parts <= 4, sum(sizes) == channels, in
 * ONE launch; sessd_nhwc_merge_nchw is its adjoint for the training step (a NULL part = a zero gradient). `sizes` and `parts` are
 * HOST arrays (of ints / device pointers). */
int sessd_nchw_split_nhwc(const float* planar, int batch, int channels, int plane, int n_parts, const int* sizes,
                          float* const* parts, sessd_stream_t stream);
int sessd_nhwc_merge_nchw(float* const* parts, int n_parts, const int* sizes, int batch, int channels, int plane, float* planar,
                          sessd_stream_t stream);
/* The same tail in TRAIN mode (both networks of the SE-SSD step, trainer_sessd.py:250-275): the two Conv2d(channels, 1, 1,
 * bias=False) weight branches, their BatchNorm2d(1) with BATCH statistics (running statistics updated in place, unbiased
 * variance, like torch), softmax and blend -- two launches forward, two backward, instead of ~30 torch / MIOpen launches.
 * x0, x1, out, grad_out, dx0, dx1: (batch, channels, plane = H * W) float32, channels % 4 == 0 (<= 1024), plane % 4 == 0.
 * w0, w1, dw0, dw1: (channels). gamma / beta / running_*: one float each (gamma, beta NULL = 1, 0; running_*: all four or none).
 * Saved by the forward for the backward: smap (2, batch * plane) = the conv outputs, stats (4) = [mean0, invstd0, mean1, invstd1].
 * dzmap (batch * plane): scratch of the backward. dgamma, dbeta (2) = gradients of [gamma0, gamma1], [beta0, beta1].
 * Workspace contract as for sessd_bn_relu_train_*: its leading 4352 bytes (arrival counters) zero on entry, zero on return.
 * Deterministic (fixed summation orders). */
size_t sessd_ssfa_fuse_train_workspace_bytes(int batch, int channels, int plane);
int sessd_ssfa_fuse_train_fwd(const float* x0, const float* x1, int batch, int channels, int plane, const float* w0,
                              const float* w1, const float* gamma0, const float* beta0, const float* gamma1, const float* beta1,
                              float eps, float momentum, float* running_mean0, float* running_var0, float* running_mean1,
                              float* running_var1, float* out, float* smap, float* stats, void* workspace,
                              size_t workspace_bytes, sessd_stream_t stream);
int sessd_ssfa_fuse_train_bwd(const float* grad_out, const float* x0, const float* x1, int batch, int channels, int plane,
                              const float* w0, const float* w1, const float* gamma0, const float* beta0, const float* gamma1,
                              const float* beta1, const float* smap, const float* stats, float* dzmap, float* dx0, float* dx1,
                              float* dw0, float* dw1, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                              sessd_stream_t stream);
/* The same tail fused with the four 1x1 heads (mg_head_sessd.py:217-230): head_w (nout, channels) row-major = the concatenated
 * conv_box | conv_cls | conv_dir | conv_iou weights, head_b (nout) or NULL, head_out (B, nout, num_pixels) planar; the blended
 * value of every channel goes into the head sums while it is in a register, `out` (the SSFA output) is written only when not
 * NULL. nout == 22, channels == 128 (the SSFA neck) or 64. */
int sessd_ssfa_fuse_head(const float* x0, const float* x1, const float* w0, const float* w1, float bn_scale0, float bn_shift0,
                         float bn_scale1, float bn_shift1, int batch, int channels, int num_pixels, float* out,
                         const float* head_w, const float* head_b, int nout, float* head_out, sessd_stream_t stream);
/* The same launch also running the score filter of MultiGroupHead.predict (mg_head_sessd.py:956-972: sigmoid(cls) >=
 * score_thresh, score *= ((iou + 1) / 2)^4) on the logits it has just produced: candidate keys (~score bits << 32 | anchor id,
 * anchor id = 2 * pixel + a) are appended to keys[b * key_cap ...], key_count[b] (device, zeroed by the caller) counts them.
 * Input of sessd_predict_fused (ext_keys / ext_key_count). keys == NULL: plain sessd_ssfa_fuse_head. */
int sessd_ssfa_fuse_head_keys(const float* x0, const float* x1, const float* w0, const float* w1, float bn_scale0,
                              float bn_shift0, float bn_scale1, float bn_shift1, int batch, int channels, int num_pixels,
                              float* out, const float* head_w, const float* head_b, int nout, float* head_out,
                              float score_thresh, unsigned long long* keys, int key_cap, int32_t* key_count,
                              sessd_stream_t stream);

/* ------------------------------------------------------------------ predict / post-processing (a11-a14)
 * replaces det3d/models/bbox_heads/mg_head_sessd.py:893-1057 (MultiGroupHead.predict / get_task_detections),
 * det3d/core/bbox/box_torch_ops.py:81-147 (second_box_decode) and :527-548 (rotate_nms), the CPU NMS of
 * det3d/ops/nms/nms_cpu.py:40-51 + nms_cpu.h:72-168 and the numba frustum test geometry.py:215-277.
 * head: (B,22,H*W) planar float32 -- ch 0..13 box codes (7 per anchor), 14..15 cls, 16..19 dir, 20..21 iou;
 * anchors (A,7) shared (anchors_per_frame = 0) or (B,A,7), A = 2*H*W; frustum (B,1,6,4,3) float64 or NULL.
 * Outputs (device): out_box (B,post,7), out_score (B,post), out_label (B,post) int32, out_count (B,). */
size_t sessd_predict_workspace_bytes(int batch, int num_anchors, int pre_max_size, int post_max_size);
int sessd_predict(const float* head, int batch, int num_pixels, const float* anchors, int anchors_per_frame,
                  const double* frustum, float score_thresh, int pre_max_size, int post_max_size, float nms_iou_thresh,
                  const float* post_center_range6, float direction_offset, float* out_box, float* out_score,
                  int32_t* out_label, int32_t* out_count, void* workspace, size_t workspace_bytes,
                  sessd_stream_t stream);
/* sessd_predict with two optional fusions (same results): ext_keys / ext_key_count (both or neither) = the score-filter keys
 * (B, 2 * num_pixels) and per-frame counts already produced by sessd_ssfa_fuse_head_keys; records / record_counts / cursor
 * (all or none) = the frame's detection record written by the call's last launch (layout and ring rule of
 * sessd_pack_detections: slot = (*cursor + b) % capacity_frames, *cursor += batch). Three launches per call with both. */
int sessd_predict_fused(const float* head, int batch, int num_pixels, const float* anchors, int anchors_per_frame,
                        const double* frustum, float score_thresh, int pre_max_size, int post_max_size,
                        float nms_iou_thresh, const float* post_center_range6, float direction_offset, float* out_box,
                        float* out_score, int32_t* out_label, int32_t* out_count, const unsigned long long* ext_keys,
                        const int32_t* ext_key_count, float* records, int32_t* record_counts, int capacity_frames,
                        int32_t* cursor, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* spconv.utils.rbbox_iou / rbbox_intersection (third-party spconv v1, imported by det3d/core/bbox/box_np_ops.py:9 for riou_cc /
 * rinter_cc :20-50): pairwise IoU (mode 0) or intersection area (mode 1) of convex quads given as corners (n,4,2) x (k,4,2);
 * pairs whose caller-supplied stand-up IoU is <= standup_thresh stay 0. standup_iou and out are (n,k) row-major. */
int sessd_quads_pairwise(int mode, const float* corners_a, int n, const float* corners_b, int k, const float* standup_iou,
                         float standup_thresh, float* out, sessd_stream_t stream);
/* DI-NMS: det3d/ops/nms/nms_cpu.h:173-384 IOU_weighted_rotate_non_max_suppression_cpu (the pybind core of
 * det3d/core/bbox/box_torch_ops.py:552-621 rotate_weighted_nms; nms_cpu.py:52-93 prepares its inputs). All arrays are device
 * pointers except sigma_dist_interval / sigma_square (host, n_interval <= 8 / n_interval - 1 used). boxes (n,7), corners (n,4,2),
 * standup_iou (n,n), scores / iou_preds (n), labels / dirs (n) int32, anchors (n, anchor_stride) or NULL with centerness_c = 0;
 * n <= 1024. Outputs (capacity n): weighted-average boxes (k,7), scores, labels, directions, kept input indices; *n_keep = k.
 * Like the reference, `thresh` of the Python signature does not exist here (the core never reads it). */
size_t sessd_di_nms_workspace_bytes(int n);
int sessd_di_nms(const float* boxes, const float* corners, const float* standup_iou, int n, const float* scores,
                 const float* iou_preds, const int* labels, const int* dirs, const float* anchors, int anchor_stride,
                 float cnt_thresh, const float* sigma_dist_interval, int n_interval, const float* sigma_square,
                 float suppressed_thresh, int centerness_c, float* boxes_ret, float* scores_ret, int* labels_ret, int* dirs_ret,
                 int* keep, int* n_keep, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* Detection records for the end-of-job gather (replaces the pickled per-rank dicts of tools/dist_test.py:150-186 /
 * det3d/torchie/trainer/utils.py:115-155): appends the `batch` frames of sessd_predict's outputs to a device ring of
 * fixed-size records (capacity_frames, post_max_size, 9) float32 [box 7 | score | label] + counts; slot = (*cursor + b) %
 * capacity_frames, then *cursor += batch (device int, so the call can sit inside a captured graph). */
int sessd_pack_detections(const float* out_box, const float* out_score, const int32_t* out_label, const int32_t* out_count,
                          int batch, int post_max_size, float* records, int32_t* record_counts, int capacity_frames,
                          int32_t* cursor, sessd_stream_t stream);
/* box_torch_ops.rotate_nms after its topk: dets (N,5) [x,y,w,l,r] sorted by descending score -> keep int32[post] */
size_t sessd_rotate_nms_workspace_bytes(int num_boxes);
int sessd_rotate_nms_sorted(const float* dets, int num_boxes, float iou_thresh, int post_max_size, int32_t* keep,
                            int32_t* num_keep, void* workspace, size_t workspace_bytes, sessd_stream_t stream);
/* det3d.ops.nms.nms.rotate_non_max_suppression_cpu (nms_cpu.h:72-168) on the device: caller-supplied corner quads (N,4,2)
 * in descending-score order; the stand-up IoU prefilter is recomputed from the corners' bounding boxes. */
int sessd_rotate_nms_corners_sorted(const float* corners, int num_boxes, float iou_thresh, int post_max_size, int32_t* keep,
                                    int32_t* num_keep, void* workspace, size_t workspace_bytes, sessd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SESSD_HIP_H */
