/*
 * mmt_hip.h -- C ABI of libmmt_hip.so: the MI355X (gfx950) BEV-fusion hot path of
 * aimotive/mm_training as hand-written HIP kernels.
 *
 * Plain pointers + sizes only; no torch / ATen types.  Every pointer is a DEVICE
 * pointer unless stated otherwise.  `stream` is a hipStream_t passed as void*
 * (NULL = the null stream).  All entry points (except mmt_voxel_pooling_plan_info, a one-time
 * read-back) are asynchronous on `stream`, never synchronise the device, never allocate and keep no pointers after returning, so
 * they can be captured into a hipGraph.
 *
 * Return value: 0 on success; MMT_ERR_* (< 0) for argument errors detected on the
 * host before anything is launched; a positive hipError_t if the launch failed.
 * mmt_last_error() returns a thread-local message for the last non-zero return.
 * (The reference prints to stderr and calls exit(-1) on a failed launch,
 * ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:51-55; this library returns
 * the error instead -- documented deviation.)
 *
 * Reference paths are relative to the aimotive/mm_training checkout.
 */
#ifndef MMT_HIP_H_
#define MMT_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: adds the cached-plan pooling, BEV warp, depth-label, CenterPoint-target and BatchNorm entry points
 * (purely additive: every version-1 symbol keeps its signature and meaning)
 * 3: adds the MMT_VP_CHUNK_POINTS field of the forward flags (bits 8-15, 0 = library default as before);
 *    mmt_bev_warp_affine_backward is bit-reproducible (gather instead of atomics) and refuses a
 *    grad_output that spans 2 GiB or more; mmt_timing_* / mmt_arm_kernel_timing (measurement support)
 * 4: mmt_hard_voxelize_mean (+ table / scratch sizes), bf16 storage entry points (*_bf16), kernel timing accepted
 *    by the lift-splat, voxelize, VFE and pillar-scatter entry points as well */
#define MMT_ABI_VERSION 15  /* 10: the plan form of the fused lift-splat forward (mmt_lss_plan_*, mmt_lss_splat_forward_plan*); additive
                             * 11: region-owner voxelizer -- mmt_voxelize_table_elems / _scratch_elems / _workspace_elems take the point
                             *     count / max_points as well; the table needs no zero fill and holds a cell directory
                             * 12: mmt_clip_adamw_step takes bf16_shadow_ptrs (may be NULL); mmt_channel_blocks_split / _gather;
                             *     fused BatchNorm also for C = a multiple of 256 up to 2048 (was: <= 1024 or 2048); mmt_bn_relu_inference;
                             *     mmt_heads_final_forward / _backward; mmt_head_loss_forward_backward
                             * 13: mmt_dcn_forward / mmt_dcn_backward (+ _supported / _workspace_bytes): the deformable convolution as implicit
                             *     GEMMs on the fp32 matrix cores, no column buffer; additive
                             * 14: mmt_lss_plan_prepare is ONE launch with a fast path for batches seen before;
                             *     mmt_depth_softmax_forward_plan_prepare (the lookup rides in the depth softmax's launch); additive
                             * 15: mmt_voxelize_fused_launch (the cells + owner passes of the voxelizer as one launch: opt-in); mmt_bev_warp_affine_backward_assign; the plan form's
                             *     job records carry a depth window per pair (3168 bytes; plan caches of ABI 14 are rebuilt); additive */

#define MMT_OK 0
#define MMT_ERR_NULL_POINTER (-1)
#define MMT_ERR_BAD_SHAPE (-2)
#define MMT_ERR_TOO_LARGE (-3)
#define MMT_ERR_BAD_FLAG (-4)
#define MMT_ERR_WORKSPACE (-5)

int mmt_abi_version(void);
const char *mmt_last_error(void);

/* Measurement support (bench.py's live roofline figure; no counterpart in the reference).  The events are attached to
 * the kernel dispatches themselves (hipExtLaunchKernel start / stop events), so the elapsed time is the kernels' own
 * duration on the device -- what rocprofv3 --kernel-trace reports -- without the dispatch latency that events recorded
 * around a launch include.  mmt_arm_kernel_timing(start, stop) applies to the NEXT call of this thread to one of
 * mmt_voxel_pooling_forward[_ex] (default SEG_GATHER launch only; the other algorithms consume and ignore it),
 * mmt_voxel_pooling_backward, mmt_lift_splat_forward / _backward, mmt_hard_voxelize[_mean], mmt_simple_vfe,
 * mmt_pillar_scatter[_nhwc][_backward] and their *_bf16 forms: start on the first kernel of the call, stop on the
 * last; the call clears it.  Other entry points leave it armed.
 * Read the result with mmt_timing_elapsed_ms after synchronising the stream. */
int mmt_timing_event_create(void **event);
int mmt_timing_event_destroy(void *event);
int mmt_timing_elapsed_ms(void *start, void *stop, float *ms);
int mmt_arm_kernel_timing(void *start, void *stop);

/* ------------------------------------------------------------------ camera half */

/* Replaces voxel_pooling_forward_kernel_launcher
 * (ops/voxel_pooling/src/voxel_pooling_forward.cpp:21-22,
 *  ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-56), i.e. what
 * voxel_pooling_forward_wrapper (voxel_pooling_forward.cpp:24-37) binds.
 * Same argument order and meaning:
 *   geom_xyz        int32 [B*P,3]   (x,y,z) integer voxel coordinates
 *   input_features  fp32  [B*P,C]   row-major
 *   output_features fp32  [B,ny,nx,C] channels-last, ACCUMULATED INTO (caller zero-fills)
 *   pos_memo        int32 [B*P,3]   rows of kept points overwritten with (b,y,x);
 *                                   rows of dropped points left untouched (caller pre-fills -1)
 * A point is kept iff 0<=x<nx && 0<=y<ny && 0<=z<nz (z only filters). */
int mmt_voxel_pooling_forward(int batch_size, int num_points, int num_channels,
                              int num_voxel_x, int num_voxel_y, int num_voxel_z,
                              const int32_t *geom_xyz, const float *input_features,
                              float *output_features, int32_t *pos_memo, void *stream);

/* Same computation with explicit algorithm / behaviour flags. */
#define MMT_VP_ALGO_AUTO 0        /* = SEG_GATHER when C % 4 == 0 and C <= 256, else LDS_ATOMIC */
#define MMT_VP_ALGO_ROW_ATOMIC 1  /* one coalesced row of global fp32 atomics per kept point */
#define MMT_VP_ALGO_LDS_ATOMIC 2  /* chunk BEV tile accumulated in LDS with ds_add_f32, then row atomics */
#define MMT_VP_ALGO_SEG_GATHER 3  /* chunk sorted by cell in LDS, rows summed in registers, one atomic row per (chunk, cell) */
#define MMT_VP_ALGO_STREAM 4      /* chunk sorted by cell, balanced stream over the sorted list, LDS row buffer */
#define MMT_VP_ALGO_MASK 0xF
#define MMT_VP_CHUNK_1024 0x20    /* SEG_GATHER: 1024 points per workgroup instead of 512 */
#define MMT_VP_WAVE_PER_SLOT 0x40 /* (ABI <= 3: the previous gather schedule, kept for A/B measurements.)  Removed in
                                     ABI 4: the bit is accepted and ignored */
#define MMT_VP_WRITE_DROPPED 0x10 /* also write (-1,-1,-1) to pos_memo rows of dropped points,
                                     so the caller need not pre-fill pos_memo */
#define MMT_VP_CHUNK_POINTS(n) ((((n) / 4) & 0xFF) << 8) /* SEG_GATHER: points per workgroup (multiple of 4,
                                     64..512); 0 = sized by the library so that the grid is a whole number of
                                     rounds of the resident workgroups.  A tuning knob: results do not depend on it
                                     beyond the fp32 summation order. */
#define MMT_VP_CHUNK_POINTS_MASK 0xFF00
int mmt_voxel_pooling_forward_ex(int batch_size, int num_points, int num_channels,
                                 int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                 const int32_t *geom_xyz, const float *input_features,
                                 float *output_features, int32_t *pos_memo, int flags,
                                 void *stream);

/* Replaces VoxelPooling.backward (ops/voxel_pooling/voxel_pooling.py:58-69), which the
 * reference runs as ATen boolean-mask + advanced-index ops:
 *   grad_in[t,:] = grad_out[b,:,y,x]  if pos_memo[t] = (b,y,x) != -1,  else 0.
 * grad_out is addressed as a [B,C,ny,nx] tensor through ELEMENT strides
 * (stride_b, stride_c, stride_y, stride_x) so any view autograd hands over works.
 * grad_in fp32 [B*P,C] is fully written (zeros for dropped points).
 * `workspace` (device, 4-byte elements, may be NULL) enables two things by its size:
 *   >= B*ny*nx*C          : a gradient with stride_c != 1 is first transposed to channels-last
 *                           there (coalesced row gather instead of a strided one);
 *   >= B*ny*nx*C + B*P    : additionally a short first pass turns pos_memo into per-point row
 *                           offsets (stored after the first B*ny*nx*C elements) and pulls the
 *                           gradient rows on-die, so the write-heavy main pass issues no HBM reads.
 * mmt_voxel_pooling_backward_workspace_elems() returns the larger figure. */
int64_t mmt_voxel_pooling_backward_workspace_elems(int batch_size, int num_points, int num_channels,
                                                   int num_voxel_x, int num_voxel_y);
int mmt_voxel_pooling_backward(int batch_size, int num_points, int num_channels,
                               int num_voxel_x, int num_voxel_y, const int32_t *pos_memo,
                               const float *grad_output, int64_t stride_b, int64_t stride_c,
                               int64_t stride_y, int64_t stride_x, float *grad_input,
                               float *workspace, int64_t workspace_elems, void *stream);

/* Replaces the quantise expression layers/backbones/lss_fpn.py:461-462
 *   ((xyz - (voxel_coord - voxel_size/2)) / voxel_size).int()
 * xyz fp32 [n,3] -> geom int32 [n,3].  voxel_coord / voxel_size are HOST pointers to 3
 * floats (the module buffers lss_fpn.py:278-285).  fp32 subtract + IEEE divide +
 * truncation toward zero (saturating, NaN -> 0), bit-exact with the reference on its device. */
int mmt_quantize_geometry(int64_t n_points, const float *xyz, const float *voxel_coord_host,
                          const float *voxel_size_host, int32_t *geom_xyz, void *stream);

/* Fused LSSFPN.get_geometry + quantise (lss_fpn.py:328-361 + :461-462): for every camera
 * (bn in [0,BN)) and frustum point s in [0,D*fH*fW): p = (u*d, v*d, d, 1),
 * xyz = (combine[bn] @ p)[:3] (fp32, k-ordered, no FMA contraction), then quantise.
 * frustum fp32 [D*fH*fW,4] (lss_fpn.py:308-326), combine fp32 [BN,4,4] = sensor2ego @ inverse(intrin).
 * Writes geom int32 [BN*D*fH*fW,3]; xyz_out (nullable) receives the fp32 points. */
int mmt_frustum_geometry(int num_cams_total, int64_t frustum_points, const float *frustum,
                         const float *combine, const float *voxel_coord_host,
                         const float *voxel_size_host, int32_t *geom_xyz, float *xyz_out,
                         void *stream);

/* Depth distribution (ABI 8).  Replaces layers/backbones/lss_fpn.py:423
 *     depth = depth_feature[:, :self.depth_channels].softmax(1)
 * and the oracle-depth overwrite of :427-438
 *     fg_mask = torch.max(depth_oracle, dim=1).values > 0.0;  depth_updated[fg_mask] = depth_oracle_flattened[fg_mask]
 * in pixel-major layout: one row of D values per pixel of the [B*N, fH, fW] feature maps -- the memory of a channels_last
 * [B*N, D, fH, fW] tensor, which is how the depth net's last 1x1 convolution leaves its logits and how the fused lift-splat
 * (MMT_LSS_PIXEL_MAJOR) reads the probabilities.
 *   logits      [pixels] rows of D, logit_row_stride elements apart (>= D: the rows may sit inside the reference's
 *               depth|context concatenation), fp32 or bf16 (logits_dtype; bf16 = the convolution ran under autocast)
 *   probs       fp32 [pixels, D]: the plain softmax (max-subtracted, expf, IEEE division) -- what `is_return_depth` returns
 *   oracle      nullable fp32 rows of D, oracle_row_stride apart: the depth labels (exps/mm_training_aim.py:259)
 *   depth_used  nullable [pixels, D] fp32 or bf16 (used_dtype): the rows the lift multiplies the context with -- the label
 *               row where it has a positive entry, the softmax elsewhere; required when oracle is given; without an oracle
 *               it is a copy of probs in used_dtype (the bf16 operand of the bf16-storage lift-splat, row g1)
 * backward: grad_logits[pix, :] = p * (g - <p, g>),  g = grad_probs + (foreground pixel ? 0 : grad_used); either gradient
 * may be NULL (= zeros); grad_logits is [pixels, D] in logits_dtype.  16-byte accesses when D % 4 == 0 and every row is
 * 16-byte aligned, element-wise otherwise; D <= 512. */
#define MMT_DTYPE_F32 0
#define MMT_DTYPE_BF16 1
int mmt_depth_softmax_forward(int64_t pixels, int D, const void *logits, int64_t logit_row_stride, int logits_dtype,
                              float *probs, const float *oracle, int64_t oracle_row_stride, void *depth_used, int used_dtype,
                              void *stream);
/* The same forward with the plan form's calibration lookup (mmt_lss_plan_prepare, below) riding in its launch (ABI 14): the
 * lookup's 1 + min(B, 8) workgroups go in front of the softmax's grid, so a step whose calibrations are known launches
 * nothing for the lookup -- as a launch of its own it costs 4.8 us on an idle card and 6-10 us inside a training step, against
 * a forward of 28 us; riding, 0.6-1.1 us on the softmax's 6.4.  A batch with calibrations to learn makes this launch as long as the
 * build (256-thread workgroups here: ~6 ms per calibration, once).  The softmax rows are those of the batch the lookup is for: pixels == B * N * fH * fW, D = the frustum's
 * depth bins.  Rows must take 16-byte pieces (D % 4 == 0, every row on a 16-byte (fp32) / 8-byte (bf16) boundary:
 * MMT_ERR_BAD_SHAPE otherwise -- make the two calls instead).  Afterwards the forward is told MMT_LSS_PLAN_PREPARED. */
int mmt_depth_softmax_forward_plan_prepare(int64_t pixels, int D, const void *logits, int64_t logit_row_stride, int logits_dtype,
                                           float *probs, const float *oracle, int64_t oracle_row_stride, void *depth_used, int used_dtype,
                                           int B, int N, int fH, int fW, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                           const float *combine, const float *frustum_u, const float *frustum_v, const float *frustum_d,
                                           const float *voxel_coord_host, const float *voxel_size_host, void *plan_cache,
                                           int64_t plan_cache_bytes, void *stream);
int mmt_depth_softmax_backward(int64_t pixels, int D, const float *probs, const float *grad_probs, const void *grad_used,
                               int used_dtype, const float *oracle, int64_t oracle_row_stride, void *grad_logits,
                               int logits_dtype, void *stream);

/* Replaces the lift + layout step layers/backbones/lss_fpn.py:441-463:
 * feats[bn,d,h,w,c] = depth[bn,d,h,w] * context[bn,c,h,w]  (fp32), written directly in the
 * [B,N,D,fH,fW,C] channels-last layout voxel_pooling consumes (no permute+contiguous copy).
 * depth fp32 [BN,D,fH*fW]; context fp32 [BN,C,fH*fW]. */
int mmt_lift_features(int num_cams_total, int D, int HW, int C, const float *depth,
                      const float *context, float *feats, void *stream);
/* and its backward: grad_depth[bn,d,s] = sum_c g[bn,d,s,c]*context[bn,c,s];
 * grad_context[bn,c,s] = sum_d g[bn,d,s,c]*depth[bn,d,s]. */
int mmt_lift_features_backward(int num_cams_total, int D, int HW, int C, const float *depth,
                               const float *context, const float *grad_feats, float *grad_depth,
                               float *grad_context, void *stream);

/* Fused lift-splat (an ADDITIONAL entry point beside the drop-in
 * pair above): voxel_pooling of features that are never materialised,
 *   out[b, y, x, :] += sum over kept points t of  depth[t] * context[pix(t), :]
 * i.e. layers/backbones/lss_fpn.py:441-464 (lift, permute+contiguous, voxel_pooling) in one
 * pass without the [B*P, C] tensor.  Points are ordered [B][N][D][HW] like geom:
 *   geom int32 [B*N*D*HW, 3]; depth fp32 [B*N, D, HW] (= one value per point, point order);
 *   context fp32 [B*N, HW, C] channels-last; out fp32 [B, ny, nx, C] accumulated into;
 *   pos_memo int32 [B*N*D*HW, 3] as in mmt_voxel_pooling_forward (flags: MMT_VP_WRITE_DROPPED).
 * Backward:  grad_depth[t] = <grad_out[cell(t), :], context[pix(t), :]>  (0 for dropped points)
 *            grad_context[pix, :] = sum_d depth[t] * grad_out[cell(t), :]
 * grad_out is indexed [B, C, ny, nx] through element strides and must be channels-last
 * (stride_c == 1); grad_depth [B*N, D, HW] and grad_context [B*N, HW, C] are fully written. */
int mmt_lift_splat_forward(int B, int N, int D, int HW, int C, int nx, int ny, int nz,
                           const int32_t *geom_xyz, const float *depth, const float *context,
                           float *output_features, int32_t *pos_memo, int flags, void *stream);
int mmt_lift_splat_backward(int B, int N, int D, int HW, int C, int nx, int ny,
                            const int32_t *pos_memo, const float *depth, const float *context,
                            const float *grad_output, int64_t stride_b, int64_t stride_c,
                            int64_t stride_y, int64_t stride_x, float *grad_depth,
                            float *grad_context, void *stream);

/* Cached pooling plan ("cached sort"; an ADDITIONAL entry point beside the
 * drop-in pair): the point -> BEV-cell assignment (voxel_pooling_forward_cuda.cu:19-29) depends only
 * on geom_xyz, i.e. on the calibration that layers/backbones/lss_fpn.py:328-361,461-462 turn into
 * indices.  While it is unchanged, sort the points by cell once and run every forward as a pure
 * segmented gather: no atomics, no geom read, no pos_memo write, bit-reproducible summation order
 * (ascending point index inside a cell, cut into items of 32 rows).
 *   plan      int32 [mmt_voxel_pooling_plan_elems(...)], 16-byte aligned, caller-owned, opaque
 *   workspace bytes [mmt_voxel_pooling_plan_workspace_bytes(...)], only needed during the build
 *   pos_memo  int32 [B*P,3] or NULL: FULLY written ((b,y,x) or (-1,-1,-1)) -- what
 *             mmt_voxel_pooling_backward consumes, so the backward of a planned forward is unchanged
 * mmt_voxel_pooling_plan_build is asynchronous like everything else.  mmt_voxel_pooling_plan_info is
 * the ONE entry point of this library that synchronises `stream`: it copies the four counts the
 * forward needs back to the host, info_host[4] = {num_items, num_kept, num_multi, num_partial}
 * (once per plan, not per step).
 * mmt_voxel_pooling_forward_planned WRITES every BEV row (zeros for empty cells; no caller memset):
 *   output_features fp32: row of cell (b,y,x) starts at out + ((b*ny+y)*nx+x) * out_row_stride
 *             (out_row_stride >= C, multiple of 4: lets the pooled map land inside a wider
 *             channels-last buffer, e.g. the camera|LiDAR concat of models/bev_depth.py:187-192)
 *   partial   fp32 [num_partial * C] scratch (may be NULL when num_multi == 0)
 * C must be a multiple of 4 and <= 256. */
int64_t mmt_voxel_pooling_plan_elems(int batch_size, int num_points, int num_voxel_x, int num_voxel_y);
int64_t mmt_voxel_pooling_plan_workspace_bytes(int batch_size, int num_points, int num_voxel_x,
                                               int num_voxel_y);
int mmt_voxel_pooling_plan_build(int batch_size, int num_points, int num_voxel_x, int num_voxel_y,
                                 int num_voxel_z, const int32_t *geom_xyz, int32_t *pos_memo,
                                 int32_t *plan, int64_t plan_elems, void *workspace,
                                 int64_t workspace_bytes, void *stream);
int mmt_voxel_pooling_plan_info(const int32_t *plan, int32_t *info_host /* HOST, 4 ints */, void *stream);
int mmt_voxel_pooling_forward_planned(int batch_size, int num_points, int num_channels, int num_voxel_x,
                                      int num_voxel_y, const int32_t *plan, int num_items, int num_multi,
                                      int num_partial, const float *input_features,
                                      float *output_features, int64_t out_row_stride, float *partial,
                                      int64_t partial_elems, void *stream);

/* BEV-augmentation warp of the pooled camera map: replaces
 * BEVDepth.bev_augment_image, models/bev_depth.py:69-84 (two kornia get_affine_matrix2d products
 * around bda_mat[:3,:3] + kornia warp_affine: bilinear, zeros padding, align_corners=True):
 *   y[b, v, u, :] = bilinear(x[b], M_b^-1 (u, v, 1)),  M_b = T(+c) bda_b[:3,:3] T(-c), c = ((W-1)/2, (H-1)/2)
 * Channels-last maps: the row of cell (b,v,u) starts at base + ((b*H+v)*W+u) * row_stride (floats,
 * >= C, multiple of 4), so input and output may be channel slices of wider buffers -- e.g. the
 * output can be the camera part of the camera|LiDAR concat buffer (models/bev_depth.py:187-192).
 * bda_mat fp32 [B, 4, 4] (last row (0,0,0,1)).  C % 4 == 0.
 * The backward ACCUMULATES into grad_input (caller zero-fills). */
int mmt_bev_warp_affine(int batch_size, int H, int W, int C, const float *bda_mat, const float *input,
                        int64_t in_row_stride, float *output, int64_t out_row_stride, void *stream);
int mmt_bev_warp_affine_backward(int batch_size, int H, int W, int C, const float *bda_mat,
                                 const float *grad_output, int64_t grad_out_row_stride,
                                 float *grad_input, int64_t grad_in_row_stride, void *stream);
/* (ABI 15) the same, but grad_input is ASSIGNED: the gather writes every row of it exactly once, so a caller that has nothing to
 * add to needs neither the zero fill nor the kernel's read of it (what autograd's backward of bev_augment_image wants).
 * Both take an input / grad_output that spans less than 2 GiB (rows are read through a buffer descriptor). */
int mmt_bev_warp_affine_backward_assign(int batch_size, int H, int W, int C, const float *bda_mat,
                                        const float *grad_output, int64_t grad_out_row_stride,
                                        float *grad_input, int64_t grad_in_row_stride, void *stream);

/* Deformable 3x3 convolution, the data-dependent halves: replaces
 * mmcv 'DCN' (DeformConv2dPack) inside DepthNet, layers/backbones/lss_fpn.py:189-197.
 * stride 1, pad 1, dilation 1, deform_groups 1.  All tensors channels-last fp32:
 *   x [B,H,W,C]; offset [B,H,W,18] = (dy,dx) per tap k=ky*3+kx;
 *   col [groups][B*H*W][9*C/groups] with K index k*(C/groups)+c_in_group.
 * The grouped GEMM between im2col and the output (and its two backward GEMMs) is plain and is
 * left to rocBLAS/hipBLASLt.  mmt_dcn_col2im ACCUMULATES into grad_x (caller zero-fills) and
 * overwrites grad_offset [B,H,W,18]. */
int mmt_dcn_im2col(int B, int H, int W, int C, int groups, const float *x, const float *offset,
                   float *col, void *stream);
int mmt_dcn_col2im(int B, int H, int W, int C, int groups, const float *x, const float *offset,
                   const float *grad_col, float *grad_x, float *grad_offset, void *stream);
/* The same backward without global atomics (ABI 4): the bilinear-corner contributions are sorted by destination pixel
 * (one LDS counting sort per image), grad_x is then a pure segmented gather -- OVERWRITTEN, the caller need not
 * zero-fill it -- and grad_offset a streaming reduction.  workspace: int32 [mmt_dcn_col2im_workspace_elems(B,H,W)]
 * (device, any contents).  Needs H*W <= 4096 and C/groups/4 a power of two <= 64; otherwise MMT_ERR_BAD_SHAPE and
 * mmt_dcn_col2im is the general form.  The order of the fp32 sums may differ between runs (as with the atomics). */
int64_t mmt_dcn_col2im_workspace_elems(int B, int H, int W);
int mmt_dcn_col2im_sorted(int B, int H, int W, int C, int groups, const float *x, const float *offset,
                          const float *grad_col, float *grad_x, float *grad_offset, int32_t *workspace,
                          int64_t workspace_elems, void *stream);

/* The whole deformable convolution as IMPLICIT GEMMs on the fp32 matrix cores (ABI 13; csrc/deform_conv_mfma.hip): no
 * [groups][B*H*W][9*C/groups] column buffer exists -- the bilinear taps are sampled into LDS tiles and multiplied there
 * (v_mfma_f32_32x32x2_f32 / _16x16x4_f32: exact fp32, a k-ordered fmaf chain).  Same operator as mmt_dcn_im2col + GEMM and
 * the GEMMs + mmt_dcn_col2im_sorted (lss_fpn.py:189-197: 3x3, stride 1, pad 1, dilation 1, deform_groups 1).  Channels-last fp32:
 *   x [B,H,W,C]; offset [B,H,W,18]; weight [O, C/groups, 3, 3] (the torch parameter as it lies); out / grad_out [B,H,W,O];
 *   grad_x [B,H,W,C] and grad_offset [B,H,W,18] are OVERWRITTEN (no zero-fill by the caller); grad_weight like weight.
 * Shapes: C/groups a multiple of 64, O/groups 64 or 128 (mmt_dcn_mfma_supported() == 1); anything else is MMT_ERR_BAD_SHAPE
 * and the im2col / col2im entry points above are the general form.  workspace: device memory, any contents, 16-byte aligned,
 * mmt_dcn_mfma_workspace_bytes(...) bytes (packed weights, the weight gradient's partial sums over pixel slices, the offset
 * gradient's partial sums over 16-channel chunks); nothing is retained between calls.
 * The forward and grad_weight / grad_offset are bit-reproducible; grad_x is summed with LDS float atomics (and, for images whose
 * rows do not fit one LDS window, global ones): its fp32 sum order may differ between runs, as with mmt_dcn_col2im[_sorted].
 * fwd_config: 0 = let the library choose the forward's workgroup shape; 1-4 = (3x2, 2x2, 4x1, 4x2) waves along (pixels x
 * output channels), 32 pixels per wave row (a tuning knob for the A/B tools). */
int mmt_dcn_mfma_supported(int B, int H, int W, int C, int O, int groups);
/* which data-gradient kernel mmt_dcn_backward runs for this shape: 0 = shape not supported, 2 = the gather form (H*W <= 768:
 * per-destination lists, no float atomics, grad_x written once), 1 = the general banded form (LDS windows summed with
 * ds_add_f32 -- 192 cycles per wave instruction on gfx950, tools/ubench/lds_atomic.hip: correct for any image, 3-4x slower
 * than the column form; callers with large images should rebuild the columns for the backward instead, as ops/deform_conv.py does) */
int mmt_dcn_backward_form(int B, int H, int W, int C, int O, int groups);
int64_t mmt_dcn_mfma_workspace_bytes(int B, int H, int W, int C, int O, int groups);
int mmt_dcn_forward(int B, int H, int W, int C, int O, int groups, const float *x, const float *offset, const float *weight,
                    float *out, void *workspace, int64_t workspace_bytes, int fwd_config, void *stream);
int mmt_dcn_backward(int B, int H, int W, int C, int O, int groups, const float *x, const float *offset, const float *weight,
                     const float *grad_out, float *grad_x, float *grad_offset, float *grad_weight, void *workspace,
                     int64_t workspace_bytes, void *stream);

/* Fused lift-splat on a camera frustum (ABI 4): the same result as mmt_lift_splat_forward for a
 * point set laid out as a frustum [B*N, D, fH, fW].  Default kernels (ABI 5): RAY WALKS -- a workgroup owns one image
 * column, its lane groups walk (depth bin, image row) and sum depth * context in registers for as long as the BEV cell
 * stays the same, one run of fp32 atomics per change of cell.  MMT_LSS_TILE_KERNELS selects the second-generation
 * frustum-tile kernels (hash + sort of a 512-point tile in LDS; less sensitive to unstructured geometry).  Either is
 * correct for any geom values; both are fast when the pixels of a column / neighbouring depth bins share cells (what a
 * pinhole frustum gives).  pos_memo may be NULL (not written).  fH <= 512, C % 4 == 0, C <= 256.
 * _bf16: depth and context stored as bf16 (C % 8 == 0), fp32 products and sums. */
#define MMT_LSS_PIXEL_MAJOR 0x100 /* flags of the mmt_lss_splat_* entry points: geom_xyz / depth / grad_depth / pos_memo are laid
                                     out PIXEL-major, [B*N, fH, fW, D(, 3)] -- the channels-last order the dense nets produce
                                     the depth distribution in -- instead of the reference's frustum order
                                     [B*N, D, fH, fW(, 3)]: a tile then reads whole 64- / 192-byte runs per pixel instead of
                                     8- / 24-byte pieces (4x fewer memory transactions: the kernels are bound by those).
                                     mmt_frustum_geometry yields that geom order when it is given the frustum permuted to
                                     [fH, fW, D, 4]. */
#define MMT_LSS_TILE_KERNELS 0x200 /* mmt_lss_splat_*: the frustum-tile kernels instead of the ray walks (A/B runs) */
#define MMT_LSS_COLUMN_BACKWARD 0x400 /* mmt_lss_splat_backward*: the COLUMN kernel on the matrix cores -- per image column and
                                        16 image rows two fp32 GEMMs (v_mfma_f32_16x16x4_f32: exact fp32) against the
                                        BEV-gradient rows of "the column's cell" per depth bin, read once per column; pixels
                                        whose own cell differs are handled one by one afterwards.  The fastest form for a
                                        level rig (every pixel of a column shares its cell: 23 us against 27 for the ray
                                        walk, 28 against 37 inside the training step), slower than the walk once more than
                                        a few per cent of the points differ (camera pitch / roll above ~1 degree).
                                        C in {64, 80, 128}; other shapes fall back to the ray walk. */
int mmt_lss_splat_forward(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y,
                          int num_voxel_z, const int32_t *geom_xyz, const float *depth, const float *context,
                          float *output_features, int32_t *pos_memo, int flags, void *stream);
int mmt_lss_splat_forward_bf16(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y,
                               int num_voxel_z, const int32_t *geom_xyz, const uint16_t *depth,
                               const uint16_t *context, float *output_features, int32_t *pos_memo, int flags,
                               void *stream);
/* Backward: the kept test is redone from geom_xyz (no pos_memo).  Ray walk: a lane group owns one pixel and walks its D
 * depth bins, context row and grad_context sum in registers, BEV-gradient rows read from L2 -- no atomics.  (Tile
 * kernels: the rows of a tile's cells in LDS, grad_context through fp32 atomics after an internal zero-fill.)
 * grad_output fp32 [B,C,ny,nx] addressed through element strides, stride_c must be 1 (channels-last).  grad_depth
 * [B*N, D, fH*fW] (fp32, or bf16 in the _bf16 form) and grad_context fp32 [B*N, fH*fW, C] are fully WRITTEN: the caller
 * need not zero-fill (and rounds grad_context to bf16 afterwards if it wants bf16).  C % 16 == 0, C <= 256, fH <= 512. */
int mmt_lss_splat_backward(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y,
                           int num_voxel_z, const int32_t *geom_xyz, const float *depth, const float *context,
                           const float *grad_output, int64_t stride_b, int64_t stride_c, int64_t stride_y,
                           int64_t stride_x, float *grad_depth, float *grad_context, int flags, void *stream);
int mmt_lss_splat_backward_bf16(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y,
                                int num_voxel_z, const int32_t *geom_xyz, const uint16_t *depth,
                                const uint16_t *context, const float *grad_output, int64_t stride_b,
                                int64_t stride_c, int64_t stride_y, int64_t stride_x, uint16_t *grad_depth,
                                float *grad_context, int flags, void *stream);

#define MMT_LSS_ZERO_OUTPUT 0x800 /* mmt_lss_splat_forward*: the entry point zero-fills output_features itself, in front of the
                                     forward kernel, with write-through stores that leave no line in the XCD L2s (the
                                     memory-side atomics that follow then do not wait for freshly written lines to be evicted,
                                     which a caller-side memset right before the launch costs them).  The caller hands over an
                                     uninitialised map; the fill is part of the call's timed kernel sequence. */

/* CAMERA FORM of the fused lift-splat (ABI 6; SURVEY section 8 rows f1 + f3): what replaces layers/backbones/lss_fpn.py:328-361
 * (get_geometry), :461-462 (quantise) and :441-464 (lift, voxel_pooling) in one kernel each way.  The geom tensor of the
 * entry points above is a pure function of B*N camera matrices and the frustum, so these kernels compute the voxel index of
 * a point themselves:
 *   p = (u[w]*d[k], v[h]*d[k], d[k], 1);  xyz = (combine[bn] @ p)[:3];  index = ((xyz - (voxel_coord - voxel_size/2)) / voxel_size).int()
 * with exactly the arithmetic of mmt_frustum_geometry (un-contracted k-ordered fp32 products and sums, correctly rounded
 * divide, truncation; csrc/mmt_camera.h) -- cells are bit-identical to mmt_frustum_geometry + the geom form on the same
 * inputs (tests/test_geometry_gpu.py::test_camera_form_*).  No geom tensor is written, read or kept: 12 bytes per point less
 * in each direction and no geometry kernel in the step.
 *   combine    device fp32 [B*N, 16]: sensor2ego @ inverse(intrin), row-major 4x4 (lss_fpn.py:339-352; rows 0..2 are used)
 *   frustum_u  device fp32 [fW] = frustum[0, 0, :, 0] (lss_fpn.py:318-320), frustum_v [fH] = frustum[0, :, 0, 1] (:321-323),
 *   frustum_d  device fp32 [D]  = frustum[:, 0, 0, 2] (:314-316); the frustum of create_frustum is the outer product of these
 *              three axes with a constant 1 in its 4th component -- a caller with any other frustum uses the geom form
 *   voxel_coord_host / voxel_size_host: 3 host floats each, as for mmt_quantize_geometry
 * Point order of depth / grad_depth / pos_memo as in the geom form (MMT_LSS_PIXEL_MAJOR or the reference's frustum order).
 * Kernels: the ray walks and the matrix-core column backward (MMT_LSS_COLUMN_BACKWARD); C in {64, 80, 128}; shapes those
 * kernels do not take (and MMT_LSS_TILE_KERNELS) return MMT_ERR_BAD_SHAPE / MMT_ERR_BAD_FLAG -- use the geom form there.
 * column_stats (backward, nullable): device uint64[2 * MMT_LSS_STATS_SLOTS], ACCUMULATED INTO by the column kernel: pair s
 * (s = workgroup index mod MMT_LSS_STATS_SLOTS, so that no address is shared by more than a few dozen workgroups) receives
 * [2s] += kept points whose cell differs from their column's (handled one by one), [2s + 1] += kept points, counted over a
 * pseudo-random 1-in-8 SAMPLE of the kernel's workgroups (workgroup index * 0x9E3779B1 >> 29 == 0); the ratio of the sums
 * over s estimates the share of points the column kernel could not take in its GEMMs.  A caller choosing between the column kernel and the ray walk reads them back lazily (no synchronisation in
 * the step); never cleared by the library. */
#define MMT_LSS_STATS_SLOTS 64
/* column_summary (nullable; device int32 [B*N, ceil(fH/16), fW, D, 2], 8-byte aligned): what the geometry of a block of 16
 * image rows of one column at one depth bin comes to -- for fixed (camera, column, bin) every coordinate is a monotone
 * function of the image row, so a block whose first and last row share their (x, y) cell shares it throughout, and its kept
 * rows are those whose z index is in range: [0] = (y << 16 | x) of that cell or -1 (outside the grid), [1] = bit i: row i's z
 * index in range, bit 16: the block does share one cell.  8 bytes per block = 0.5 byte per point (a geom tensor: 12).
 *   forward, flags without MMT_LSS_SUMMARY_CACHED: the kernel computes the geometry and WRITES the summary on the way;
 *   forward with MMT_LSS_SUMMARY_CACHED: the summary was written by an earlier call with the SAME combine / frustum / grid
 *            (an unchanged calibration): the kernel READS it instead of computing;
 *   backward: READ when given (the forward's), otherwise the geometry is computed again.
 * Blocks without bit 16 (a camera that is not level: some column straddles a cell border) are evaluated row by row from the
 * matrices by whichever kernel meets them, so combine / frustum_* are always required.  Needs nx, ny < 32768. */
#define MMT_LSS_SUMMARY_CACHED 0x1000
/* exclusive_cache (ABI 7; forward, nullable; device int32, 16-byte aligned, exclusive_cache_bytes =
 * mmt_lss_exclusive_cache_bytes(N, nx, ny, slots) bytes, ZERO-INITIALISED ONCE by the caller and then left to the library):
 * a persistent per-calibration memory of which BEV cells receive ONE run of the forward only.  Such a cell (43 % of the runs
 * on the cfg4 rig: far cells seen by one column of one camera) needs no atomic add -- the run is stored into the zero-filled
 * row -- and the forward is bound by the memory-side atomic units.  The library learns them on the device, per sample: the
 * table is direct-mapped by a 64-bit hash of the sample's N matrices (compared bit for bit on a hit); the first call that
 * presents a calibration claims its slot, the second MARKs (every run stores its id into the cell's state), the third
 * VERIFIES (a run that finds another id stores -1), and from the fourth on runs into single-run cells leave as plain stores.
 * Every workgroup of the forward looks its sample up itself (two rounds of loads behind its geometry phase); what a call
 * decides for the next one is posted in a mailbox that the zero-fill kernel in front of the next forward commits: no host
 * involvement, no synchronisation, no workgroup waits for another, capturable in a graph.  A change of the launch shape, of
 * the grid or of the frustum axes' contents empties the table.  Used with MMT_LSS_ZERO_OUTPUT by the register-walk forward
 * (fH <= 16, C <= 80, D < 160; mmt_lss_last_kernel_family reports MMT_LSS_FAMILY_REGISTER | _EXCLUSIVE); ignored otherwise.
 * Results are those of the call without a cache up to the order of additions (a stored sum is a sum added to 0).
 * Calls that share a cache must be ordered on one stream.  Limits: 65536 slots, samples of at most 8 cameras, the first 8
 * samples of a call (further samples run without it).  Header words [8 + b] / [24 + b] tell what the last call found for
 * sample b: its slot and the stage (0 miss, 1 mark, 2 verify, 3 use). */
int64_t mmt_lss_exclusive_cache_bytes(int N, int num_voxel_x, int num_voxel_y, int slots);   /* 0: bad arguments */
int mmt_lss_splat_forward_cam(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                              const float *combine, const float *frustum_u, const float *frustum_v, const float *frustum_d,
                              const float *voxel_coord_host, const float *voxel_size_host, const float *depth,
                              const float *context, float *output_features, int32_t *pos_memo, int32_t *column_summary,
                              int32_t *exclusive_cache, int64_t exclusive_cache_bytes, int flags, void *stream);
int mmt_lss_splat_forward_cam_bf16(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                   const float *combine, const float *frustum_u, const float *frustum_v,
                                   const float *frustum_d, const float *voxel_coord_host, const float *voxel_size_host,
                                   const uint16_t *depth, const uint16_t *context, float *output_features,
                                   int32_t *pos_memo, int32_t *column_summary, int32_t *exclusive_cache,
                                   int64_t exclusive_cache_bytes, int flags, void *stream);
int mmt_lss_splat_backward_cam(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                               const float *combine, const float *frustum_u, const float *frustum_v, const float *frustum_d,
                               const float *voxel_coord_host, const float *voxel_size_host, const float *depth,
                               const float *context, const float *grad_output, int64_t stride_b, int64_t stride_c,
                               int64_t stride_y, int64_t stride_x, float *grad_depth, float *grad_context,
                               const int32_t *column_summary, uint64_t *column_stats, int flags, void *stream);
int mmt_lss_splat_backward_cam_bf16(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y,
                                    int num_voxel_z, const float *combine, const float *frustum_u, const float *frustum_v,
                                    const float *frustum_d, const float *voxel_coord_host, const float *voxel_size_host,
                                    const uint16_t *depth, const uint16_t *context, const float *grad_output,
                                    int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x,
                                    uint16_t *grad_depth, float *grad_context, const int32_t *column_summary,
                                    uint64_t *column_stats, int flags, void *stream);
/* Which kernel family the process's last mmt_lss_splat_forward* (backward = 0) / mmt_lss_splat_backward* (backward = 1)
 * call launched (process-wide, not per thread: an autograd backward runs on the engine's thread): MMT_LSS_FAMILY_* below, ORed with MMT_LSS_FAMILY_CAMERA for the camera form; 0 before the first call.  The
 * entry points pick a family from the shape (LDS budgets, channel count), so a fallback is visible here. */
#define MMT_LSS_FAMILY_RAY 1
#define MMT_LSS_FAMILY_TILE 2
#define MMT_LSS_FAMILY_COLUMN 3
#define MMT_LSS_FAMILY_CAMERA 0x10
#define MMT_LSS_FAMILY_REGISTER 0x20 /* forward, ORed to MMT_LSS_FAMILY_RAY: the register walk (columns of up to 16 rows, C <= 80, D < 160) */
#define MMT_LSS_FAMILY_EXCLUSIVE 0x40 /* forward: the call used an exclusive-cell cache */
#define MMT_LSS_FAMILY_BLOCK 0x80 /* forward, camera form, ORed to MMT_LSS_FAMILY_RAY: the block walk (columns of more than 16 rows, long rays, C = 128) */
int mmt_lss_last_kernel_family(int backward);
/* 1 when mmt_lss_splat_forward_cam AND mmt_lss_splat_backward_cam both take this shape, 0 otherwise (use the geom form). */
int mmt_lss_camera_form_supported(int B, int N, int D, int fH, int fW, int C);
/* 1 when mmt_lss_splat_forward_cam[_bf16] of this shape uses an exclusive_cache it is handed (ABI 8): a caller allocates the
 * cache (mmt_lss_exclusive_cache_bytes) only for such shapes; every other shape ignores the argument. */
int mmt_lss_exclusive_cache_used(int B, int N, int D, int fH, int fW, int C);

/* PLAN FORM of the fused lift-splat forward (ABI 10; SURVEY section 8 rows f1 + f3; replaces layers/backbones/lss_fpn.py:328-361,
 * :461-462 and :441-464 like the camera form, same operands, same cells bit for bit): the OUTPUT-stationary forward.  Which
 * frustum points feed a BEV cell depends on the calibration (a sample's N matrices, the frustum axes, the grid) only, so the
 * library learns it once per calibration, on the device, into a PLAN kept in the caller's plan cache: runs (up to 4
 * consecutive depth bins of a 16-row block of one image column that share a cell, with 16-bit row masks) grouped into jobs
 * (a contiguous range of the cells of an 8 x 8 BEV tile, at most 96 runs).  The forward gives a job to a workgroup: lane
 * groups sum depth * context per run in registers (the pair's 16 context rows loaded once), one partial row per run into
 * LDS, then every cell's partial rows are summed in plan order and STORED.  Against the camera form's ray walks: no zero
 * fill of the map, no global atomics, every output element written exactly once, results bit-identical from call to call
 * (the sums are ordered by camera, column, row block, bin).
 *   plan_cache  device memory, 256-byte aligned, plan_cache_bytes = mmt_lss_plan_cache_bytes(..., slots) bytes with
 *               slots >= B; owned by the caller for good, contents owned by the library (no initialisation needed; calls that
 *               share a cache must be ordered on one stream).  A slot holds one calibration (a few MB: cfg4 3.9 MB); the least
 *               recently used one is replaced.  A change of the launch shape, the grid or the frustum axes' contents empties it.
 *   mmt_lss_plan_prepare  looks every sample of the batch up (64-bit hash of its matrices, then bit for bit) and learns the
 *               calibrations it does not know (ONE launch: workgroup 0 looks up, the others wait for its word and build; a batch seen before --
 *               one of the last four, bit for bit -- takes one round of loads: 4.8 us on an idle card).  It depends on `combine` only:
 *               call it as early in the step as the matrices exist and pass MMT_LSS_PLAN_PREPARED to the forward -- or leave
 *               the flag out and the forward does it itself in front of its kernel.
 *   column_summary (forward, nullable): the batch's column summary [B*N, ceil(fH/16), fW, D, 2] as the camera form defines
 *               it, WRITTEN (copied from the slots) for mmt_lss_splat_backward_cam.
 * depth / context as in the camera form with MMT_LSS_PIXEL_MAJOR (required).  B <= 64, N <= 16, C in {64, 80, 128},
 * 4 <= D <= 2047, fH <= 512, N * fW <= 65535, nx, ny <= 32767.  A calibration whose plan does not fit its slot (more than
 * 2 * N * ceil(fH/16) * fW * D runs: cameras rolled so far that most 16-row blocks straddle cells) is served by the same kernel's brute-force path from the slot's summary: exact, deterministic, slow
 * (~0.5 ms per sample) -- header word 10 of the cache counts such samples; use the camera form for such a rig. */
#define MMT_LSS_PLAN_PREPARED 0x2000 /* mmt_lss_splat_forward_plan*: mmt_lss_plan_prepare ran for this batch on this stream since the last forward */
#define MMT_LSS_PLAN_BRUTE 0x4000    /* mmt_lss_splat_forward_plan*: every sample through the brute-force path (tests) */
#define MMT_LSS_FAMILY_PLAN 4        /* mmt_lss_last_kernel_family(0) & 0xF: the plan form */
int mmt_lss_plan_supported(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y, int num_voxel_z);
int64_t mmt_lss_plan_cache_bytes(int N, int D, int fH, int fW, int num_voxel_x, int num_voxel_y, int slots);   /* 0: bad arguments */
int mmt_lss_plan_prepare(int B, int N, int D, int fH, int fW, int num_voxel_x, int num_voxel_y, int num_voxel_z, const float *combine,
                         const float *frustum_u, const float *frustum_v, const float *frustum_d, const float *voxel_coord_host,
                         const float *voxel_size_host, void *plan_cache, int64_t plan_cache_bytes, void *stream);
int mmt_lss_splat_forward_plan(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                               const float *combine, const float *frustum_u, const float *frustum_v, const float *frustum_d,
                               const float *voxel_coord_host, const float *voxel_size_host, const float *depth, const float *context,
                               float *output_features, int32_t *column_summary, void *plan_cache, int64_t plan_cache_bytes, int flags,
                               void *stream);
int mmt_lss_splat_forward_plan_bf16(int B, int N, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                    const float *combine, const float *frustum_u, const float *frustum_v, const float *frustum_d,
                                    const float *voxel_coord_host, const float *voxel_size_host, const uint16_t *depth,
                                    const uint16_t *context, float *output_features, int32_t *column_summary, void *plan_cache,
                                    int64_t plan_cache_bytes, int flags, void *stream);
/* Counters of a plan cache (device -> host copy of 8 words: the call synchronises `stream`): [0] sample-calls that found their
 * plan, [1] calibrations learnt, [2] sample-calls served by the brute-force path, [3] times the table was emptied, [4] calls,
 * [5] slots. */
int mmt_lss_plan_cache_counters(const void *plan_cache, int64_t plan_cache_bytes, int64_t *counters_host /* 8 */, void *stream);
/* Where things are in a plan cache of plan_cache_bytes bytes for this shape (diagnostics and tests; 12 host words): [0] slots,
 * [1] byte offset of slot 0, [2] bytes per slot, [3] / [4] offsets of a slot's column summary / job records, [5] offset of the
 * per-sample verdicts (int32 x 4 each: slot, jobs, state 1 = planned / 2 = brute force, representative sample), [6] job capacity,
 * [7] run capacity, [8] bytes per job record, [9] verdict entries, [10] strips (N * ceil(fH/16) * fW), [11] tiles. */
int mmt_lss_plan_cache_layout(int N, int D, int fH, int fW, int num_voxel_x, int num_voxel_y, int64_t plan_cache_bytes, int64_t *layout_host);

/* ------------------------------------------------- bf16 feature storage
 * BASELINE configs[4] names bf16.  The reference has no behaviour for it -- its extension takes data_ptr<float>()
 * only (ops/voxel_pooling/src/voxel_pooling_forward.cpp:28-31; exps/conf_aim.py:30 "16 does not work yet") -- so
 * SURVEY 5.6 defines it: bf16 STORAGE of the big operands, fp32 accumulation, parity against the fp32 oracle on the
 * up-cast inputs.  uint16_t = raw bf16 bits.  Same argument meaning, ownership and flags as the fp32 entry points:
 *   mmt_voxel_pooling_forward_bf16   input_features bf16 [B*P,C] (C % 8 == 0, C <= 512), output_features fp32
 *                                    (accumulated into); flags: MMT_VP_WRITE_DROPPED / MMT_VP_CHUNK_POINTS only
 *   mmt_voxel_pooling_backward_bf16  grad_output fp32 (any strides), grad_input bf16 [B*P,C]: the gathered fp32 row
 *                                    rounded to nearest even (exact: the backward is a copy)
 *   mmt_lift_features_bf16           depth / context fp32 (they come from the fp32 dense nets) -> feats bf16
 *                                    [BN,D,HW,C] = bf16(depth * context); _backward_bf16: grad_feats bf16 ->
 *                                    grad_depth / grad_context fp32 (fp32 sums)
 *   mmt_lift_splat_forward_bf16      depth bf16 [B*N,D,HW], context bf16 [B*N,HW,C] -> BEV fp32 (products and sums
 *                                    fp32); _backward_bf16: grad_out fp32 -> grad_depth / grad_context bf16 */
int mmt_voxel_pooling_forward_bf16(int batch_size, int num_points, int num_channels, int num_voxel_x,
                                   int num_voxel_y, int num_voxel_z, const int32_t *geom_xyz,
                                   const uint16_t *input_features, float *output_features, int32_t *pos_memo,
                                   int flags, void *stream);
int mmt_voxel_pooling_backward_bf16(int batch_size, int num_points, int num_channels, int num_voxel_x,
                                    int num_voxel_y, const int32_t *pos_memo, const float *grad_output,
                                    int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x,
                                    uint16_t *grad_input, float *workspace, int64_t workspace_elems, void *stream);
int mmt_lift_features_bf16(int BN, int D, int HW, int C, const float *depth, const float *context,
                           uint16_t *feats, void *stream);
int mmt_lift_features_backward_bf16(int BN, int D, int HW, int C, const float *depth, const float *context,
                                    const uint16_t *grad_feats, float *grad_depth, float *grad_context,
                                    void *stream);
int mmt_lift_splat_forward_bf16(int B, int N, int D, int HW, int C, int num_voxel_x, int num_voxel_y,
                                int num_voxel_z, const int32_t *geom_xyz, const uint16_t *depth,
                                const uint16_t *context, float *output_features, int32_t *pos_memo, int flags,
                                void *stream);
int mmt_lift_splat_backward_bf16(int B, int N, int D, int HW, int C, int num_voxel_x, int num_voxel_y,
                                 const int32_t *pos_memo, const uint16_t *depth, const uint16_t *context,
                                 const float *grad_output, int64_t stride_b, int64_t stride_c, int64_t stride_y,
                                 int64_t stride_x, uint16_t *grad_depth, uint16_t *grad_context, void *stream);

/* ------------------------------------------------------------------- LiDAR half */

/* Replaces mmcv-full 1.7.0 ops.Voxelization (hard, deterministic) as called per sample by
 * mmdet3d MVXTwoStageDetector.voxelize (call site models/bev_depth.py:181; parameters
 * exps/conf_aim.py:16-18,194-197), batched over B samples in one call.
 *   points        fp32 [sum Ni, F]   samples concatenated
 *   point_offsets int32 [B+1]        DEVICE prefix offsets of the samples in `points`
 *   voxel_size / range_min  HOST float[3];  grid HOST int[3] (x,y,z)
 * Outputs are laid out with a fixed per-sample capacity so nothing is sized by a device count:
 *   voxels       fp32  [B*max_voxels, max_points, F]  sample b starts at row b*max_voxels
 *   coors        int32 [B*max_voxels, 4] = (b,z,y,x)
 *   num_points   int32 [B*max_voxels]
 *   voxel_count  int32 [B]   number of voxels of each sample (M_b <= max_voxels)
 * Voxels are numbered in order of their first point; <= max_points points per voxel,
 * first come first kept; voxels beyond max_voxels (in first-point order) are dropped.
 * workspace: int32, at least mmt_voxelize_workspace_elems(...) elements (device), any contents (table + scratch of
 * mmt_hard_voxelize_mean below in one buffer: every word the kernels read is written by them first). */
int64_t mmt_voxelize_workspace_elems(int batch_size, int64_t total_points, const int32_t *grid_host, int max_points);
/* The cells pass and the region-owner pass of the voxelizer as ONE launch (the owners wait, inside the launch, for their
 * sample's cells workgroups: write-through stores + a token word per workgroup, no fences): two launches per voxelization
 * instead of three, same outputs bit for bit.  OFF by default -- on MI355X the hand-off inside the launch costs more than
 * the kernel boundary it replaces (24.3 against 23.1 us at 4 x 40 k points).  on: 0 / 1 sets it for every
 * later call of this process (also: MMT_VOX_FUSED=1 in the environment), < 0 only asks; returns the previous setting.
 * Replaces nothing in the reference (models/bev_depth.py:181 calls mmcv's op); a measurement knob. */
int mmt_voxelize_fused_launch(int on);
int mmt_hard_voxelize(int batch_size, int64_t total_points, int num_features,
                      const float *points, const int32_t *point_offsets,
                      const float *voxel_size_host, const float *range_min_host,
                      const int32_t *grid_host, int max_points, int max_voxels, float *voxels,
                      int32_t *coors, int32_t *num_points, int32_t *voxel_count,
                      int32_t *workspace, void *stream);

/* The same voxelization fused with the HardSimpleVFE mean (models/bev_depth.py:181-182 in one call; SURVEY
 * section 8 row a10: "fuse into a9's epilogue"), no clearing pass, no memory atomics (ABI 11: the region-owner form --
 * a workgroup per run of 1024..4096 consecutive cells settles them in LDS; lidar_voxelize.hip):
 *   table    int32 [mmt_voxelize_table_elems(B, grid, total_points)], 8-byte aligned, any contents.  The call leaves the
 *            cloud's CELL DIRECTORY in it (an occupancy bit per cell, block ordinals, the head point of every occupied
 *            cell, the voxel id of every head) -- what mmt_pillar_scatter_nhwc_table[_strided] read instead of a
 *            cell -> row map.  One table serves one stream at a time.
 *   scratch  int32 [mmt_voxelize_scratch_elems(B, grid, total_points, max_points)], any contents.
 *   voxels   may be NULL: the padded [B*max_voxels, max_points, F] tensor is then not materialised
 *            (4*T*F bytes per voxel less traffic) -- what LidarEncoder.forward_bev does.
 *   mean     fp32 [B*max_voxels, num_features] or NULL: sum over the voxel's points (slot order) of the first
 *            num_features columns / num_points; rows past voxel_count[b] are written as zeros.
 * Other arguments and outputs as mmt_hard_voxelize.  total_points < 2^24, max_points <= 127. */
int64_t mmt_voxelize_table_elems(int batch_size, const int32_t *grid_host, int64_t total_points);
int64_t mmt_voxelize_scratch_elems(int batch_size, const int32_t *grid_host, int64_t total_points, int max_points);
int mmt_hard_voxelize_mean(int batch_size, int64_t total_points, int num_features,
                           const float *points, const int32_t *point_offsets,
                           const float *voxel_size_host, const float *range_min_host,
                           const int32_t *grid_host, int max_points, int max_voxels,
                           int vfe_num_features, float *voxels, int32_t *coors, int32_t *num_points,
                           int32_t *voxel_count, float *mean, int32_t *table, int32_t *scratch,
                           void *stream);

/* Compacts the fixed-capacity outputs above into the dense (concatenated) tensors the
 * reference returns: rows [sum_{b'<b} M_b', ...) <- sample b's first M_b rows.
 * dst_offsets int32 [B+1] device prefix sums of voxel_count. */
int mmt_compact_voxels(int batch_size, int max_voxels, int row_elems_voxels,
                       const int32_t *voxel_count, const int32_t *dst_offsets,
                       const float *voxels, const int32_t *coors, const int32_t *num_points,
                       float *voxels_out, int32_t *coors_out, int32_t *num_points_out,
                       void *stream);

/* Replaces mmdet3d 1.0.0rc4 HardSimpleVFE(num_features) (models/bev_depth.py:182;
 * exps/conf_aim.py:198-201): out[m,k] = sum_t voxels[m,t,k] / num_points[m], k < num_features,
 * summed in slot order. */
int mmt_simple_vfe(int64_t num_voxels, int max_points, int F, int num_features,
                   const float *voxels, const int32_t *num_points, float *out, void *stream);

/* Replaces mmdet3d PointPillarsScatter behind pts_middle_encoder(voxel_feats, coors, batch_size)
 * (models/bev_depth.py:183): canvas[b,:,y,x] = feats[m,:] for coors[m]=(b,z,y,x), zeros
 * elsewhere; canvas fp32 [B,C,ny,nx] is fully written (memset folded in).
 * workspace: int32 [B*ny*nx] (device) cell -> row map.  Duplicate cells: highest row wins. */
int mmt_pillar_scatter(int64_t num_voxels, int C, int batch_size, int ny, int nx,
                       const float *feats, const int32_t *coors, float *canvas,
                       int32_t *workspace, void *stream);
/* grad_feats[m,:] = grad_canvas[b,:,y,x] for rows owning their cell (else 0);
 * uses the cell->row map left in `workspace` by mmt_pillar_scatter. */
int mmt_pillar_scatter_backward(int64_t num_voxels, int C, int batch_size, int ny, int nx,
                                const float *grad_canvas, const int32_t *coors,
                                const int32_t *workspace, float *grad_feats, void *stream);

/* Channels-last variant of the pillar scatter: canvas fp32 [B, ny, nx, C] (the memory the channels_last BEV
 * convolutions consume; returned to torch as a [B, C, ny, nx] view), same last-writer semantics, same
 * workspace (cell -> row map).  A cell is one contiguous C-float row: the scatter writes and the backward
 * gathers whole rows instead of C values ny*nx floats apart.  C % 4 == 0. */
int mmt_pillar_scatter_nhwc(int64_t num_voxels, int C, int batch_size, int ny, int nx, const float *voxel_features,
                            const int32_t *coors, float *canvas, int32_t *workspace, void *stream);
int mmt_pillar_scatter_nhwc_backward(int64_t num_voxels, int C, int batch_size, int ny, int nx,
                                     const float *grad_canvas, const int32_t *coors, const int32_t *workspace,
                                     float *grad_feats, void *stream);

/* Pillar scatter straight from the voxelizer's table (ABI 4; what LidarEncoder.forward_bev runs): for the fixed-capacity
 * rows of an mmt_hard_voxelize_mean call on `table` -- distinct cells by construction -- the canvas pass walks the cell
 * directory that call left (occupancy bit -> ordinal -> head point -> voxel id) instead of building a cell -> row map: one
 * kernel, no fill, no atomics; a table no voxelization has written yields a canvas of NaN.  Must run on the same stream after that voxelization and before the next one on the table; needs a
 * single z layer and (ny, nx) = the voxel grid's (y, x); feats fp32 [B*max_voxels, C], canvas fp32 [B, ny, nx, C] fully written.
 * _unique_backward: grad_feats[m,:] = grad_canvas[cell(coors[m]),:] for rows whose coors are valid, 0 otherwise -- no map. */
int mmt_pillar_scatter_nhwc_table(int C, int batch_size, int ny, int nx, int max_voxels, const float *voxel_features,
                                  const int32_t *table, float *canvas, void *stream);
int mmt_pillar_scatter_nhwc_unique_backward(int64_t num_voxels, int C, int batch_size, int ny, int nx,
                                            const float *grad_canvas, const int32_t *coors, float *grad_feats,
                                            void *stream);

/* Pillar scatter at the resolution the fusion layer consumes (ABI 8; replaces models/bev_depth.py:183 + :188-190 +
 * the LiDAR half of the torch.cat at :192).  The reference scatters the full-resolution canvas [B, C, ny, nx] and
 * nearest-resizes it onto the camera BEV grid: for an integer ratio (stride_y, stride_x) = (ny / out_h, nx / out_w) torch's
 * 'nearest' reads canvas cell (i * stride_y, j * stride_x) for output cell (i, j) and nothing else.  These entry points
 * write exactly those cells: out[b, i, j, 0:C] = canvas[b, :, i * stride_y, j * stride_x], rows `out_row_stride` floats
 * apart (a channels-last [B, ny / stride_y, nx / stride_x, out_row_stride] buffer; `out` already points at the LiDAR
 * channel offset of the camera|LiDAR concat buffer).  Bit-identical to mmt_pillar_scatter_nhwc[_table] followed by
 * [..., ::stride_y, ::stride_x]; strides must divide the grid, C % 4 == 0, out_row_stride % 4 == 0, 16-byte aligned.
 *   _table_strided   : the rows of the last mmt_hard_voxelize_mean call on `table` (see mmt_pillar_scatter_nhwc_table).
 *   _strided         : any (feats, coors) rows, last-writer rule; workspace int32 [B * (ny/stride_y) * (nx/stride_x)].
 *   _strided_backward: grad_feats[m,:] = grad_out[b, y / stride_y, x / stride_x, 0:C] for rows on a sampled cell
 *                      (y % stride_y == 0 && x % stride_x == 0) that own it, zeros otherwise; `workspace` = the map the
 *                      _strided forward left, or NULL when the rows own distinct cells (the table form).  grad_out rows are
 *                      grad_row_stride floats apart (the gradient of the concat buffer, advanced to the channel offset). */
int mmt_pillar_scatter_nhwc_table_strided(int C, int batch_size, int ny, int nx, int max_voxels, int stride_y, int stride_x,
                                          const float *voxel_features, const int32_t *table, float *out,
                                          int64_t out_row_stride, void *stream);
int mmt_pillar_scatter_nhwc_strided(int64_t num_voxels, int C, int batch_size, int ny, int nx, int stride_y, int stride_x,
                                    const float *voxel_features, const int32_t *coors, float *out, int64_t out_row_stride,
                                    int32_t *workspace, void *stream);
int mmt_pillar_scatter_nhwc_strided_backward(int64_t num_voxels, int C, int batch_size, int ny, int nx, int stride_y,
                                             int stride_x, const float *grad_out, int64_t grad_row_stride,
                                             const int32_t *coors, const int32_t *workspace, float *grad_feats,
                                             void *stream);

/* --------------------------------------------------------- per-step label generation */

/* LiDAR depth supervision of the camera branch: replaces
 * get_depth_labels / get_depth_image / get_downsampled_gt_depth,
 * exps/mm_training_aim.py:114-163,180-215 (B x N_cam Python loop of projections, a dense
 * H x W depth image per camera, block minimum, depth-bin index, one-hot).
 *   points        fp32 [sum Ni, F]  all samples' points concatenated (columns x,y,z first)
 *   point_offsets int32 [B+1]       first row of each sample (DEVICE); max_points >= max Ni (host)
 *   extrinsics    fp32 [B, N, 4, 4] ego -> camera;  intrinsics fp32 [B, N, 4, 4]
 *   bda_inv       fp32 [B, 3, 3]    inverse of the BEV-augmentation rotation (:129-131)
 *   workspace     int32 [mmt_depth_labels_workspace_elems(...)] per-cell minimum depth (float bits)
 *   depth_bin     int32 [B*N*fH*fW] or NULL: bin index, 0 = no / out-of-range depth (:207-212)
 *   onehot        fp32  [B*N*fH*fW, D] or NULL: F.one_hot(bin, D).float() (:213-214)
 * A point counts iff depth > 1, 1 < u < W-1, 1 < v < H-1 (:150-155); its cell is
 * (int(v)/downsample, int(u)/downsample); the cell keeps the minimum depth (:199-204).
 * Two points on one PIXEL: the reference keeps the one written last, here the smaller one. */
int64_t mmt_depth_labels_workspace_elems(int batch_size, int num_cams, int img_h, int img_w, int downsample);
int mmt_depth_labels(int batch_size, int num_cams, int point_features, int max_points, int img_h,
                     int img_w, int downsample, float d_lo, float d_step, int depth_channels,
                     const float *points, const int32_t *point_offsets, const float *extrinsics,
                     const float *intrinsics, const float *bda_inv, int32_t *workspace,
                     int64_t workspace_elems, int32_t *depth_bin, float *onehot, void *stream);

/* The same with the per-camera horizontal flip of augment_images (exps/mm_training_aim.py:89-112) folded into the label
 * write (ABI 8): flipped = device bytes [B*N], non-zero = this camera's label map is written mirrored along w (what
 * kornia.hflip of the [D, fH, fW] label image gives); NULL = mmt_depth_labels. */
int mmt_depth_labels_flipped(int batch_size, int num_cams, int point_features, int max_points, int img_h,
                             int img_w, int downsample, float d_lo, float d_step, int depth_channels,
                             const float *points, const int32_t *point_offsets, const float *extrinsics,
                             const float *intrinsics, const float *bda_inv, int32_t *workspace,
                             int64_t workspace_elems, int32_t *depth_bin, float *onehot, const uint8_t *flipped, void *stream);

/* Image augmentation of the training step (ABI 8): exps/mm_training_aim.py:89-112
 * (augment_images: per camera, with probability 1/2, kornia hflip of the image and of its depth-label map -- the reference
 * stacks per-image Python lists) and :510-512 (normalize_images: torchvision Normalize((0.485, 0.456, 0.406), (0.229, 0.224,
 * 0.225)) of sweep_imgs[:, :, :, :3] / 255).  The flags are one device byte per camera, drawn on the host like the reference
 * draws them (np.random.uniform(size = b*s*n) > 0.5) and also handed to LSSFPN as mats['flipped'].
 *   mmt_hflip: out[i, r, w, :] = in[i, r, flipped[i / group] ? W-1-w : w, :] for a contiguous fp32 [n, rows, W, elems]
 *     tensor (label maps [B*N, fH, fW, D]: group 1; NCHW images viewed as [B*N*3, H, W, 1]: group 3).  Out of place.
 *   mmt_normalize_flip_images: out[i, c, h, w] = (images[i, c, h, ws] * scale - mean[c]) / std[c], ws = W-1-w for a flipped
 *     camera, c < 3 of channels_in >= 3 -- one pass instead of three element-wise kernels, a flip and a select; with
 *     channels_last the result is stored as [n, H, W, 3] (the memory of a channels_last [n, 3, H, W] tensor, what the first
 *     convolution reads).  `scale` multiplies (ATen evaluates `/ 255.` as a multiplication by the fp32 reciprocal on the GPU);
 *     mean / std: 3 host floats each; flipped nullable. */
int mmt_hflip(int64_t n, int group, int rows, int W, int elems, const float *in, const uint8_t *flipped, float *out,
              void *stream);
int mmt_normalize_flip_images(int64_t n_images, int channels_in, int H, int W, const float *images, float scale,
                              const float *mean_host, const float *std_host, const uint8_t *flipped, float *out,
                              int channels_last, void *stream);

/* CenterPoint training targets: replaces BEVDepthHead.get_targets_single,
 * layers/heads/bev_depth_head.py:113-254 (Python loop over tasks and boxes around mmdet3d's
 * gaussian_radius / draw_heatmap_gaussian).  HOST arrays: class_begin / class_count [num_tasks]
 * (first label and number of classes of each task) and the four arrays of num_tasks DEVICE
 * pointers.  Per task t (all FULLY written):
 *   heatmaps[t]   fp32  [B, class_count[t], fy, fx]  max-combined Gaussians (:212)
 *   anno_boxes[t] fp32  [B, max_objs, 10]  (dx, dy, z, log w, log l, log h, sin, cos, vx, vy) (:220-234)
 *   inds[t]       int64 [B, max_objs]      y * fx + x of the centre cell (:218);  masks[t] uint8 [B, max_objs]
 * boxes fp32 [sum K, 9] (x,y,z,w,l,h,yaw,vx,vy), labels int32 [sum K], box_offsets int32 [B+1]
 * (DEVICE), max_boxes >= max K (host).  Box k of a sample fills slot k of the task owning its
 * class (the reference packs each task's boxes densely; its loss only sums masked slots, so the
 * slot order is immaterial); boxes beyond max_objs are ignored (:171). */
int mmt_centerpoint_targets(int batch_size, int num_tasks, const int32_t *class_begin,
                            const int32_t *class_count, int max_objs, int max_boxes, int fx, int fy,
                            float range_x0, float range_y0, float voxel_x, float voxel_y,
                            int out_size_factor, float gaussian_overlap, int min_radius, int norm_bbox,
                            const float *boxes, const int32_t *labels, const int32_t *box_offsets,
                            float *const *heatmaps, float *const *anno_boxes, int64_t *const *inds,
                            uint8_t *const *masks, void *stream);

/* ------------------------------------------------- dense-net glue: BatchNorm + add + ReLU
 * Training-mode BatchNorm2d fused with the optional residual add and ReLU that follow it in the
 * ResNet / FPN / DepthNet blocks (mmcv / mmdet modules in the reference: layers/backbones/lss_fpn.py,
 * layers/heads/bev_depth_head.py), for channels-last fp32 activations viewed as [R = N*H*W, C]:
 *   y = relu?( (x - mean) * rstd * weight + bias  [+ residual] ),  batch statistics over R,
 *   running_mean / running_var updated with `momentum` (unbiased variance), torch semantics.
 * 3 + 5 streaming passes instead of 5 + 8 for MIOpen BN + ATen add / relu.
 *   workspace fp32 [mmt_bn_workspace_elems(C)] scratch (per-workgroup partial sums; may be shared by
 *             all layers that run on one stream)
 *   save      fp32 [4*C]: mean | rstd | scale | shift, written by forward, read by backward
 * C % 4 == 0 and (C <= 1024 or C == 2048).  Backward: grad_residual (if has_residual) is the
 * ReLU-masked grad_y; y is needed only when relu && has_residual (the mask comes from the output). */
int64_t mmt_bn_workspace_elems(int C);
int mmt_bn_relu_forward(int64_t R, int C, const float *x, const float *residual, const float *weight,
                        const float *bias, float *running_mean, float *running_var, float momentum,
                        float eps, int relu, float *workspace, float *save, float *y, void *stream);
int mmt_bn_relu_backward(int64_t R, int C, const float *x, const float *y, const float *grad_y,
                         const float *save, int relu, int has_residual, float *workspace, float *grad_x,
                         float *grad_residual, float *grad_weight, float *grad_bias, void *stream);
/* ABI 9: the same with the ACTIVATIONS (x, residual, y and their gradients) stored as act_dtype = MMT_DTYPE_F32 or MMT_DTYPE_BF16
 * (8-byte aligned rows then); statistics, weight / bias, running statistics, save and every sum stay fp32 -- the arithmetic of
 * batch_norm inside a torch.autocast(bf16) region (BASELINE configs[4]), which otherwise runs through MIOpen's NHWC batch norm. */
int mmt_bn_relu_forward_ex(int64_t R, int C, const void *x, const void *residual, const float *weight,
                           const float *bias, float *running_mean, float *running_var, float momentum,
                           float eps, int relu, float *workspace, float *save, void *y, int act_dtype, void *stream);
int mmt_bn_relu_backward_ex(int64_t R, int C, const void *x, const void *y, const void *grad_y,
                            const float *save, int relu, int has_residual, float *workspace, void *grad_x,
                            void *grad_residual, float *grad_weight, float *grad_bias, int act_dtype, void *stream);
/* ABI 12: eval-mode BatchNorm (+ residual) (+ ReLU) in one pass: y = relu?(x * w / sqrt(running_var + eps) + (b - running_mean * ..)
 * [+ residual]); nothing is reduced, nothing updated.  The reference's image backbone is built with frozen_stages=0
 * (exps/conf_aim.py:57: mmdet ResNet keeps conv1 + norm1 without gradients and norm1 in eval mode while training).
 * workspace: fp32 [2 * C] scratch. */
int mmt_bn_relu_inference(int64_t R, int C, const void *x, const void *residual, const float *weight, const float *bias,
                          const float *running_mean, const float *running_var, float eps, int relu, float *workspace,
                          void *y, int act_dtype, void *stream);
/* ABI 11: the same with a SECOND and a THIRD gradient of y (nullable; the third only with the second), added to grad_y on load.
 * The output of a residual block is read twice -- by the next block's first convolution and as its identity (or through its
 * downsample convolution), a stage's output a third time by the neck -- and autograd would add the gradients in passes of its own
 * (three streams over the activation each) before this backward reads the sum; `ops/bn_relu.py::bn_act(..., fork=2|3)` hands the
 * output out as aliases instead and receives their gradients here.  grad_y_row_stride: 0 (dense rows of C) or the distance in
 * elements between the rows of grad_y when it is a channel slice of a wider channels-last tensor (the gradient of one input of a
 * torch.cat: read in place instead of through a .contiguous() copy); >= C, a multiple of 4. */
int mmt_bn_relu_backward_ex2(int64_t R, int C, const void *x, const void *y, const void *grad_y, const void *grad_y2,
                             const void *grad_y3, int64_t grad_y_row_stride, const float *save, int relu, int has_residual, float *workspace, void *grad_x,
                             void *grad_residual, float *grad_weight, float *grad_bias, int act_dtype, void *stream);

/* ------------------------------------------------------------------- optimizer step (row a13) */

/* Gradient clipping + AdamW over every parameter in two launches (ABI 11).  Replaces, per step of the reference's harness
 * (exps/mm_training_aim.py:575-608: Lightning's gradient_clip_val = 2 = torch.nn.utils.clip_grad_norm_, then torch.optim.AdamW):
 * the multi-tensor norm, the multiply of all gradients by the clip coefficient and the fused AdamW kernels -- the multiply's read +
 * write of every gradient disappears into the update (the gradient is scaled on load; the gradients themselves are left unscaled).
 *   chunk_tensor int32 [num_chunks], chunk_offset int64 [num_chunks]: tensor index and first element of every chunk of
 *     chunk_elems elements (a multiple of 4; a tensor's last chunk may be shorter); all DEVICE arrays
 *   param_ptrs / grad_ptrs / exp_avg_ptrs / exp_avg_sq_ptrs int64 [T] DEVICE arrays of device addresses (fp32 tensors, dense), numel int64 [T]
 *   bf16_shadow_ptrs int64 [T] or NULL: per tensor the address of a bf16 copy of the parameter (same memory order; 0 = none), rewritten
 *     with the updated values (round to nearest even) -- the autocast convolutions' weights without a cast kernel per layer and step
 *   step >= 1: the update's number (bias corrections 1 - beta^step); max_norm <= 0: no clipping (and no norm launch)
 *   partials fp32 [num_chunks] scratch (max_norm > 0); norm_out fp32 [2] or NULL: total gradient norm, clip coefficient applied
 * Arithmetic as torch's fused kernel (ADAMW, amsgrad off, maximize off): double products with lr / betas / weight decay / eps,
 * fp32 state; clip coefficient = min(1, max_norm / (norm + 1e-6)). */
int mmt_clip_adamw_step(int num_chunks, int chunk_elems, const int32_t *chunk_tensor, const int64_t *chunk_offset,
                        const int64_t *param_ptrs, const int64_t *grad_ptrs, const int64_t *exp_avg_ptrs,
                        const int64_t *exp_avg_sq_ptrs, const int64_t *bf16_shadow_ptrs, const int64_t *numel, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int64_t step, float max_norm, float *partials, float *norm_out, void *stream);

/* out = inputs[0] + ... + inputs[n-1] (n <= 32 dense fp32 tensors of numel elements; inputs_host: a HOST array of device pointers,
 * passed to the kernel by value), summed in argument order, one pass: the gradient of a tensor that n branches read
 * (layers/heads/bev_depth_head.py: the 24 branches of the CenterPoint head) without autograd's n - 1 accumulation passes. */
int mmt_add_n(int n, const void *const *inputs_host, int64_t numel, float *out, void *stream);

/* A channels-last activation [rows, n * W] (n blocks of W channels per pixel; block_bytes = W * element size, a multiple of 16)
 * <-> n dense [rows, W] tensors (parts_host: a HOST array of n <= 32 device pointers, 16-byte aligned), one pass either way.  The
 * task heads' 24 first convolutions (the reference's 24 ConvModules on one shared map, bev_depth_head.py SeparateHead) run as ONE
 * 64 -> 24 x 64 convolution and ONE BatchNorm; `split` hands every final convolution its 64 channels as a dense tensor, `gather`
 * puts the 24 gradients that come back side by side again. */
int mmt_channel_blocks_split(int64_t rows, int n, int block_bytes, const void *wide, void *const *parts_host, void *stream);
int mmt_channel_blocks_gather(int64_t rows, int n, int block_bytes, const void *const *parts_host, void *wide, void *stream);

/* The task heads' FINAL convolutions in one launch per direction (ABI 12; csrc/thin_conv.hip).  The reference's SeparateHead ends every
 * branch with Conv2d(64, classes, 3, padding=1, bias=True), classes = 1..3 (layers/heads/bev_depth_head.py; mmdet3d CenterHead): NB
 * branches (<= 32), branch j reading channels [64 j, 64 j + 64) of ONE channels-last map z [B, H, W, NB * 64] and writing k_host[j]
 * (1..4) channels of ONE channels-last map out [B, H, W, KT], KT = sum of k_host, its channels at the running sum of k_host.
 *   weight fp32 [KT][9][64] (= the memory of a channels-last [KT, 64, 3, 3] tensor: the branches' weights one after the other), bias
 *   fp32 [KT] or NULL; z / out / grad_out / grad_z: act_dtype (fp32 or bf16; fp32 arithmetic)
 *   backward: grad_z (nullable) [B, H, W, NB * 64]; grad_weight [KT][9][64] + grad_bias [KT] + workspace (nullable together;
 *   workspace: mmt_heads_final_workspace_elems(B, H, NB) floats) -- per-workgroup partial sums added in a fixed order, no atomics */
/* The CenterPoint head's loss and its gradient on the fused heads' one output map, two launches (ABI 12; csrc/head_loss.hip; the
 * reference: layers/heads/bev_depth_head.py:256-312 = mmdet GaussianFocalLoss(alpha 2, gamma 4) on clip_sigmoid(heatmap, 1e-4) +
 * L1 on the boxes gathered at `inds`, weighted by mask * !isnan(target) * code_weights * loss_bbox.loss_weight, each divided by the
 * task's normaliser).  map [B, H, W, 11 T] (act_dtype): per task reg 0-1, height 2, dim 3-5, rot 6-7, vel 8-9, heatmap 10.
 *   heatmaps_host / anno_host / inds_host / masks_host: HOST arrays of T device pointers (fp32 [B, H, W] / fp32 [B, M, 10] /
 *   int64 [B, M] / uint8 [B, M]); normalisers fp32 [2 T] on the device (positives per task, masked slots per task, before the clamps
 *   to >= 1 and >= 1e-4); code_weights fp32 [10]
 *   grad_map fp32 [B, H, W, 11 T]: d loss / d map, written whole; partials fp32 [mmt_head_loss_partials(..)]: the loss = their sum */
int mmt_head_loss_partials(int B, int H, int W, int T, int M);
int mmt_head_loss_forward_backward(int B, int H, int W, int T, int M, const void *map, const void *const *heatmaps_host,
                                   const void *const *anno_host, const void *const *inds_host, const void *const *masks_host,
                                   const float *normalisers, const float *code_weights, float box_weight, float *grad_map,
                                   float *partials, int act_dtype, void *stream);

int64_t mmt_heads_final_workspace_elems(int B, int H, int NB);
int mmt_heads_final_forward(int B, int H, int W, int NB, const unsigned char *k_host, const void *z, const float *weight,
                            const float *bias, void *out, int act_dtype, void *stream);
int mmt_heads_final_backward(int B, int H, int W, int NB, const unsigned char *k_host, const void *z, const float *weight,
                             const void *grad_out, void *grad_z, float *grad_weight, float *grad_bias, float *workspace,
                             int act_dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMT_HIP_H_ */
