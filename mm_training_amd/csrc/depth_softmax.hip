// Depth distribution of the camera branch for MI355X (gfx950, wave64) -- SURVEY section 8 row a8, first clause:
//   layers/backbones/lss_fpn.py:423      depth = depth_feature[:, :self.depth_channels].softmax(1)
//   layers/backbones/lss_fpn.py:427-438  depth_updated[fg_mask] = depth_oracle_flattened[fg_mask]   (oracle-depth overwrite)
// in the layout the fused lift-splat reads.  The logits leave the depth net's last 1x1 convolution as a channels_last
// [B*N, D, fH, fW] tensor, i.e. one contiguous run of D floats per pixel (possibly inside a wider row: the reference's
// DepthNet concatenates depth | context on the channel axis); the fused op reads the probabilities pixel-major
// [B*N, fH, fW, D], which is the channels_last view of the [B*N, D, fH, fW] tensor `is_return_depth` hands to the depth loss.
// ATen's softmax over a channel slice of that tensor (cunn_SpatialSoftMaxForward / Backward) took 75 + 66 us per step at
// BASELINE configs[3] for 2 x 7.6 MB and wrote NCHW, which the fused op then transposed again.  Here a lane group of 16
// lanes owns one pixel: the D logits are read once as 16-byte pieces (D = 112: 28 pieces over 16 lanes), maximum and sum
// meet through four DPP steps inside the row of 16 lanes (no LDS, no barrier), and the row leaves as 16-byte stores.
// The oracle overwrite is folded in: a pixel whose label row has a positive entry (torch.max(oracle, 1).values > 0) takes
// the label row as the depth the lift uses, every other pixel its own softmax; `probs` (what the loss sees) is the plain
// softmax either way, exactly the two tensors the reference ends up with.
// Backward: grad_logits = p * (g - <p, g>) with g = grad_probs + (foreground ? 0 : grad_used) -- torch's softmax backward on
// the sum of the two consumers' gradients, the second masked like the `where` it came through.
#include "mmt_common.h"

namespace {

#include "depth_softmax_body.h"

constexpr int kBlock = 256;
constexpr int kGroupsPerBlock = kBlock / kGroup;

// LT: logits type, UT: depth_used type (float / bf16_t); NV pieces of VEC elements per lane cover D <= 16 * NV * VEC
template <typename LT, typename UT, int VEC, int NV>
__global__ __launch_bounds__(kBlock) void depth_softmax_fwd(SoftmaxArgs a) {
    softmax_fwd_rows<LT, UT, VEC, NV>(a, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, kBlock);
}

template <typename LT, typename UT, int VEC, int NV>
__global__ __launch_bounds__(kBlock) void depth_softmax_bwd(SoftmaxArgs a) {
    const int grp = threadIdx.x / kGroup, lane = threadIdx.x % kGroup;
    const int D = a.D;
    for (int64_t pix = (int64_t)blockIdx.x * kGroupsPerBlock + grp; pix < a.pixels; pix += (int64_t)gridDim.x * kGroupsPerBlock) {
        float p[NV * VEC], g[NV * VEC], t[NV * VEC];
        const float *prow = a.probs + pix * D;
#pragma unroll
        for (int k = 0; k < NV; ++k) load_piece<float, VEC>(prow, (k * kGroup + lane) * VEC, D, 0.f, p + k * VEC);
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) g[i] = 0.f;
        if (a.grad_probs != nullptr) {
            const float *grow = a.grad_probs + pix * D;
#pragma unroll
            for (int k = 0; k < NV; ++k) load_piece<float, VEC>(grow, (k * kGroup + lane) * VEC, D, 0.f, g + k * VEC);
        }
        if (a.grad_used != nullptr) {
            bool fg = false;
            if (a.oracle != nullptr) {
                const float *orow = a.oracle + pix * a.oracle_stride;
#pragma unroll
                for (int k = 0; k < NV; ++k) load_piece<float, VEC>(orow, (k * kGroup + lane) * VEC, D, 0.f, t + k * VEC);
                float om = t[0];
#pragma unroll
                for (int i = 1; i < NV * VEC; ++i) om = fmaxf(om, t[i]);
                fg = row16_reduce<true>(om) > 0.f;
            }
            if (!fg) {          // a foreground pixel's depth_used is the label row: nothing flows to the logits from there
                const UT *urow = static_cast<const UT *>(a.grad_used) + pix * D;
#pragma unroll
                for (int k = 0; k < NV; ++k) load_piece<UT, VEC>(urow, (k * kGroup + lane) * VEC, D, 0.f, t + k * VEC);
#pragma unroll
                for (int i = 0; i < NV * VEC; ++i) g[i] += t[i];
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) dot += p[i] * g[i];
        dot = row16_reduce<false>(dot);
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) g[i] = (g[i] - dot) * p[i];     // ATen: (grad - sum(grad * output)) * output
        LT *lrow = static_cast<LT *>(a.grad_logits) + pix * D;
#pragma unroll
        for (int k = 0; k < NV; ++k) store_piece<LT, VEC>(lrow, (k * kGroup + lane) * VEC, D, g + k * VEC);
    }
}

template <typename LT, typename UT, bool FWD>
void launch_width(const SoftmaxArgs &a, bool vec4, mmt::TimedSeq &seq, hipStream_t st) {
    const dim3 grid(mmt::stream_grid(a.pixels * kGroup, kBlock, 256 * 32)), block(kBlock);
    const int pieces = (int)mmt::ceil_div(a.D, kGroup * (vec4 ? 4 : 1));     // per lane
#define MMT_SM_LAUNCH(V, N)                                                                        \
    {                                                                                              \
        if (FWD) seq.launch(true, depth_softmax_fwd<LT, UT, V, N>, grid, block, 0, st, a);         \
        else seq.launch(true, depth_softmax_bwd<LT, UT, V, N>, grid, block, 0, st, a);             \
    }
    if (vec4) {
        if (pieces <= 2) MMT_SM_LAUNCH(4, 2)
        else if (pieces <= 4) MMT_SM_LAUNCH(4, 4)
        else MMT_SM_LAUNCH(4, 8)
    } else {
        if (pieces <= 8) MMT_SM_LAUNCH(1, 8)
        else if (pieces <= 16) MMT_SM_LAUNCH(1, 16)
        else MMT_SM_LAUNCH(1, 32)
    }
#undef MMT_SM_LAUNCH
}

template <bool FWD>
void launch_types(const SoftmaxArgs &a, bool logits_bf16, bool used_bf16, bool vec4, mmt::TimedSeq &seq, hipStream_t st) {
    if (logits_bf16) {
        if (used_bf16) launch_width<bf16_t, bf16_t, FWD>(a, vec4, seq, st);
        else launch_width<bf16_t, float, FWD>(a, vec4, seq, st);
    } else {
        if (used_bf16) launch_width<float, bf16_t, FWD>(a, vec4, seq, st);
        else launch_width<float, float, FWD>(a, vec4, seq, st);
    }
}

}  // namespace

extern "C" int mmt_depth_softmax_forward(int64_t pixels, int D, const void *logits, int64_t logit_row_stride, int logits_dtype,
                                         float *probs, const float *oracle, int64_t oracle_row_stride, void *depth_used,
                                         int used_dtype, void *stream) {
    if (int rc = softmax_common_check("depth_softmax_forward", pixels, D, logits_dtype, used_dtype)) return rc;
    if (pixels == 0) return MMT_OK;
    MMT_REQUIRE_PTR(logits);
    MMT_REQUIRE_PTR(probs);
    if (logit_row_stride < D || (oracle != nullptr && oracle_row_stride < D))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "depth_softmax_forward: row strides must be >= D");
    if (oracle != nullptr && depth_used == nullptr)
        return mmt::fail(MMT_ERR_NULL_POINTER, "depth_softmax_forward: an oracle needs a depth_used output");
    const bool lb = logits_dtype == MMT_DTYPE_BF16, ub = used_dtype == MMT_DTYPE_BF16;
    const bool vec4 = D % 4 == 0 && softmax_aligned(logits, logit_row_stride, lb ? 2 : 4) && softmax_aligned(probs, D, 4) &&
                      softmax_aligned(oracle, oracle_row_stride, 4) && softmax_aligned(depth_used, D, ub ? 2 : 4);
    SoftmaxArgs a{pixels, D, logits, logit_row_stride, probs, oracle, oracle_row_stride, depth_used, nullptr, nullptr, nullptr};
    mmt::TimedSeq seq;
    launch_types<true>(a, lb, ub, vec4, seq, (hipStream_t)stream);
    return mmt::check_launch("depth_softmax_forward");
}

extern "C" int mmt_depth_softmax_backward(int64_t pixels, int D, const float *probs, const float *grad_probs, const void *grad_used,
                                          int used_dtype, const float *oracle, int64_t oracle_row_stride, void *grad_logits,
                                          int logits_dtype, void *stream) {
    if (int rc = softmax_common_check("depth_softmax_backward", pixels, D, logits_dtype, used_dtype)) return rc;
    if (pixels == 0) return MMT_OK;
    MMT_REQUIRE_PTR(probs);
    MMT_REQUIRE_PTR(grad_logits);
    if (oracle != nullptr && oracle_row_stride < D) return mmt::fail(MMT_ERR_BAD_SHAPE, "depth_softmax_backward: row strides must be >= D");
    const bool lb = logits_dtype == MMT_DTYPE_BF16, ub = used_dtype == MMT_DTYPE_BF16;
    const bool vec4 = D % 4 == 0 && softmax_aligned(probs, D, 4) && softmax_aligned(grad_probs, D, 4) && softmax_aligned(grad_used, D, ub ? 2 : 4) &&
                      softmax_aligned(oracle, oracle_row_stride, 4) && softmax_aligned(grad_logits, D, lb ? 2 : 4);
    SoftmaxArgs a{pixels, D, nullptr, 0, const_cast<float *>(probs), oracle, oracle_row_stride, nullptr, grad_probs, grad_used, grad_logits};
    mmt::TimedSeq seq;
    launch_types<false>(a, lb, ub, vec4, seq, (hipStream_t)stream);
    return mmt::check_launch("depth_softmax_backward");
}
