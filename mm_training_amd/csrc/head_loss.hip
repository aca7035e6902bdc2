// The CenterPoint head's loss on the fused heads' ONE output map (layers/heads/bev_depth_head.py::loss; the reference:
// bev_depth_head.py:256-312 = mmdet GaussianFocalLoss on clip_sigmoid(heatmap) + masked, code-weighted L1 on the boxes gathered at
// `inds`).  As torch ops that is ~150 launches on 64 K-element tensors per step, forward and backward, between the end of the forward
// pass and the start of the backward pass where nothing overlaps them.  Here: the loss AND its gradient w.r.t. the map in two
// launches (the gradient is d loss / d map for an upstream gradient of 1; autograd scales it).
//   map  [B, H, W, KT] fp32 / bf16, KT = 11 * T: per task t the channels 11 t + (reg 0-1, height 2, dim 3-5, rot 6-7, vel 8-9, heatmap 10)
//   head_loss_dense : per pixel and task  hm = clamp(sigmoid(logit), 1e-4, 1 - 1e-4)
//                     l = [-log(hm + 1e-12) (1 - hm)^2 (target == 1)  -  log(1 - hm + 1e-12) hm^2 (1 - target)^4] / cls_norm[t]
//                     writes the pixel's whole gradient row (zeros on the box channels), one loss partial per workgroup
//   head_loss_boxes : per (task, sample, slot, code)  |map[b, ind, 11 t + c] - target| * mask * !isnan(target) * code_weight[c]
//                     * box_weight / box_norm[t]; the gradient (its sign) is ADDED to the row (fp32 atomic: two boxes may share a pixel)
// Partials are summed by the caller (a fixed order: the same bits every run, up to the order of two boxes on one pixel).
#include "mmt_common.h"

namespace {

constexpr int kMaxTasks = 8;
constexpr int kCodes = 10;
constexpr int kTaskChannels = 11;

struct HeadLossArgs {
    const void *map;
    float *grad;                    // [B, H, W, KT] fp32
    const float *heat[kMaxTasks];   // [B, H, W] targets
    const float *anno[kMaxTasks];   // [B, M, 10]
    const int64_t *ind[kMaxTasks];  // [B, M]
    const unsigned char *mask[kMaxTasks];
    const float *norm;              // [2 T]: positives per task, masked slots per task (before the clamps)
    const float *code_w;            // [10]
    float box_weight;
    float *partial;                 // [dense blocks + box blocks]
    int64_t pixels;                 // B * H * W
    int HW, T, M, B, dense_blocks;
};

template <typename AT> __device__ __forceinline__ float ldm(const void *p, int64_t i);
template <> __device__ __forceinline__ float ldm<float>(const void *p, int64_t i) { return static_cast<const float *>(p)[i]; }
template <> __device__ __forceinline__ float ldm<bf16_t>(const void *p, int64_t i) {
    return __uint_as_float((unsigned)static_cast<const bf16_t *>(p)[i] << 16);
}

__device__ __forceinline__ float block_sum(float v, float *red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename AT>
__global__ __launch_bounds__(256) void head_loss_dense(HeadLossArgs a) {
    __shared__ float red[4];
    const int KT = a.T * kTaskChannels;
    float loss = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.pixels; p += (int64_t)gridDim.x * 256) {
        float *g = a.grad + p * KT;
        for (int t = 0; t < a.T; ++t) {
            const float x = ldm<AT>(a.map, p * KT + t * kTaskChannels + 10);
            const float tg = a.heat[t][p];
            const float inv = 1.f / fmaxf(a.norm[t], 1.f);
            const float s = 1.f / (1.f + expf(-x));
            const bool inside = s >= 1e-4f && s <= 1.f - 1e-4f;          // (torch.clamp hands the gradient on at the bounds themselves)
            const float hm = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
            const float pos = tg == 1.f ? 1.f : 0.f;
            const float omt = 1.f - tg, negw = (omt * omt) * (omt * omt);
            const float lp = logf(hm + 1e-12f), ln = logf(1.f - hm + 1e-12f), omh = 1.f - hm;
            loss += (-lp * (omh * omh) * pos - ln * (hm * hm) * negw) * inv;
            const float dpos = pos * (-(omh * omh) / (hm + 1e-12f) + 2.f * omh * lp);
            const float dneg = negw * ((hm * hm) / (1.f - hm + 1e-12f) - 2.f * hm * ln);
            const float gl = inside ? (dpos + dneg) * (s * (1.f - s)) * inv : 0.f;
#pragma unroll
            for (int c = 0; c < 10; ++c) g[t * kTaskChannels + c] = 0.f;
            g[t * kTaskChannels + 10] = gl;
        }
    }
    const float total = block_sum(loss, red);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = total;
}

template <typename AT>
__global__ __launch_bounds__(256) void head_loss_boxes(HeadLossArgs a) {
    __shared__ float red[4];
    const int KT = a.T * kTaskChannels;
    const int64_t n = (int64_t)a.T * a.B * a.M * kCodes;
    float loss = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % kCodes);
        int64_t r = i / kCodes;
        const int o = (int)(r % a.M); r /= a.M;
        const int b = (int)(r % a.B), t = (int)(r / a.B);
        const int64_t slot = (int64_t)b * a.M + o;
        if (!a.mask[t][slot]) continue;
        const float tg = a.anno[t][slot * kCodes + c];
        if (tg != tg) continue;                                                 // (NaN targets carry no weight)
        int64_t pix = a.ind[t][slot];
        pix = pix < 0 ? 0 : (pix >= a.HW ? a.HW - 1 : pix);
        const int64_t e = ((int64_t)b * a.HW + pix) * KT + t * kTaskChannels + c;
        const float w = a.code_w[c] * a.box_weight / fmaxf(a.norm[a.T + t], 1e-4f);
        const float d = ldm<AT>(a.map, e) - tg;
        loss += fabsf(d) * w;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        if (sg != 0.f && w != 0.f) atomicAdd(a.grad + e, sg * w);
    }
    const float total = block_sum(loss, red);
    if (threadIdx.x == 0) a.partial[a.dense_blocks + blockIdx.x] = total;
}

}  // namespace

extern "C" int mmt_head_loss_partials(int B, int H, int W, int T, int M) {
    if (B < 1 || H < 1 || W < 1 || T < 1 || T > kMaxTasks || M < 0) return -1;
    const int dense = mmt::stream_grid((int64_t)B * H * W, 256, 1024);
    const int boxes = M > 0 ? mmt::stream_grid((int64_t)T * B * M * kCodes, 256, 256) : 0;
    return dense + boxes;
}

extern "C" int mmt_head_loss_forward_backward(int B, int H, int W, int T, int M, const void *map, const void *const *heatmaps_host,
                                              const void *const *anno_host, const void *const *inds_host, const void *const *masks_host,
                                              const float *normalisers, const float *code_weights, float box_weight, float *grad_map,
                                              float *partials, int act_dtype, void *stream) {
    MMT_REQUIRE_PTR(map);
    MMT_REQUIRE_PTR(heatmaps_host);
    MMT_REQUIRE_PTR(normalisers);
    MMT_REQUIRE_PTR(code_weights);
    MMT_REQUIRE_PTR(grad_map);
    MMT_REQUIRE_PTR(partials);
    if (B < 1 || H < 1 || W < 1 || T < 1 || T > kMaxTasks || M < 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "head_loss: B, H, W >= 1, 1 <= tasks <= %d, slots >= 0 (B=%d H=%d W=%d T=%d M=%d)", kMaxTasks, B, H, W, T, M);
    if (act_dtype != MMT_DTYPE_F32 && act_dtype != MMT_DTYPE_BF16) return mmt::fail(MMT_ERR_BAD_FLAG, "head_loss: unknown activation dtype %d", act_dtype);
    if (M > 0 && !(anno_host && inds_host && masks_host)) return mmt::fail(MMT_ERR_NULL_POINTER, "head_loss: box targets are NULL");
    HeadLossArgs a = {};
    a.map = map; a.grad = grad_map; a.norm = normalisers; a.code_w = code_weights; a.box_weight = box_weight; a.partial = partials;
    a.pixels = (int64_t)B * H * W; a.HW = H * W; a.T = T; a.M = M; a.B = B;
    for (int t = 0; t < T; ++t) {
        if (!heatmaps_host[t] || (M > 0 && !(anno_host[t] && inds_host[t] && masks_host[t]))) return mmt::fail(MMT_ERR_NULL_POINTER, "head_loss: a target of task %d is NULL", t);
        a.heat[t] = static_cast<const float *>(heatmaps_host[t]);
        if (M > 0) {
            a.anno[t] = static_cast<const float *>(anno_host[t]);
            a.ind[t] = static_cast<const int64_t *>(inds_host[t]);
            a.mask[t] = static_cast<const unsigned char *>(masks_host[t]);
        }
    }
    a.dense_blocks = mmt::stream_grid(a.pixels, 256, 1024);
    hipStream_t st = (hipStream_t)stream;
    if (act_dtype == MMT_DTYPE_F32) hipLaunchKernelGGL(head_loss_dense<float>, dim3(a.dense_blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(head_loss_dense<bf16_t>, dim3(a.dense_blocks), dim3(256), 0, st, a);
    if (int rc = mmt::check_launch("head_loss(dense)")) return rc;
    if (M > 0) {
        const int blocks = mmt::stream_grid((int64_t)T * B * M * kCodes, 256, 256);
        if (act_dtype == MMT_DTYPE_F32) hipLaunchKernelGGL(head_loss_boxes<float>, dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(head_loss_boxes<bf16_t>, dim3(blocks), dim3(256), 0, st, a);
        if (int rc = mmt::check_launch("head_loss(boxes)")) return rc;
    }
    return MMT_OK;
}
