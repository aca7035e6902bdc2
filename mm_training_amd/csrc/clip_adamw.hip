// Gradient clipping + AdamW as two launches over every parameter of the model (SURVEY section 8 row a13:
// exps/mm_training_aim.py:575-608 -- `gradient_clip_val=2` and `torch.optim.AdamW(lr=1e-3/64*bs, weight_decay=1e-7)`).
//
// torch runs this as clip_grad_norm_ (a multi-tensor norm, a handful of scalar kernels, a multi-tensor multiply over all gradients)
// followed by the fused AdamW: at BASELINE configs[3] (83.9 M parameters) 0.14 + 0.19 + 0.75 ms per step, of which the multiply is a
// whole read + write of the gradients that only scales them for the kernel that follows.  Here:
//   1. opt_sumsq : a workgroup per 64 K-element chunk of a gradient: sum of squares -> one partial per chunk;
//   2. opt_adamw : a workgroup per chunk: adds the partials up in a fixed order (1 300 floats from L2: every workgroup for itself,
//                  same value everywhere), total norm -> clip coefficient = min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.
//                  clip_grad_norm_), then the AdamW update of its chunk with the gradient scaled on load -- torch's fused kernel's
//                  arithmetic (fused_adam_utils.cuh, ADAMW, amsgrad off): decay, first and second moment, bias corrections
//                  from the step count, one division.
// The tensors are named by device arrays of pointers (parameters and moments: built once; gradients: uploaded per step, autograd
// allocates them afresh) and a chunk table; 16-byte accesses where a chunk's four pointers allow it.
#include "mmt_common.h"

namespace {

constexpr int kOptBlock = 256;

struct OptArgs {
    const int64_t *p, *g, *m, *v;     // [T] device pointers as integers
    const int64_t *shadow;            // [T] or NULL: bf16 copies of the parameters (0: none for this tensor), rewritten with the update
    const int64_t *numel;             // [T]
    const int32_t *chunk_tensor;      // [NC]
    const int64_t *chunk_off;         // [NC] first element of the chunk inside its tensor
    int chunk_elems, NC;
    float *partials;                  // [NC]
    float *norm_out;                  // [2] total norm, clip coefficient (written by workgroup 0 of opt_adamw); may be NULL
    double lr_d, beta1_d, beta2_d, eps_d, wd_d;
    float bc1, bc2_sqrt, max_norm;
};

__device__ __forceinline__ float block_sum(float v, float *lds) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kOptBlock / 64; ++w) t += lds[w];      // (fixed order: the same bits in every workgroup)
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(kOptBlock) void opt_sumsq(OptArgs a) {
    __shared__ float lds[kOptBlock / 64];
    const int t = a.chunk_tensor[blockIdx.x];
    const int64_t off = a.chunk_off[blockIdx.x];
    const int64_t left = a.numel[t] - off;
    const int n = (int)(left < a.chunk_elems ? left : a.chunk_elems);
    const float *g = reinterpret_cast<const float *>(a.g[t]) + off;
    float s = 0.f;
    if (((uintptr_t)g & 15) == 0) {
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        for (int i = threadIdx.x; i < (n >> 2); i += kOptBlock) { const float4 x = g4[i]; s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }
        for (int i = (n & ~3) + threadIdx.x; i < n; i += kOptBlock) s += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < n; i += kOptBlock) s += g[i] * g[i];
    }
    s = block_sum(s, lds);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
}

// torch's fused kernel (fused_adam_utils.cuh::adam_math, ADAMW, amsgrad off) keeps lr, betas, weight decay and eps as doubles: the
// products with them are formed in double and rounded when assigned to the fp32 value -- mirrored here so that the two agree to the bit
// wherever the gradients do
__device__ __forceinline__ void adamw1(float &p, float g, float &m, float &v, const OptArgs &a, float step_size) {
    const double lr = (double)a.lr_d, wd = (double)a.wd_d, b1 = (double)a.beta1_d, b2 = (double)a.beta2_d;
    p = (float)((double)p - lr * wd * (double)p);
    m = (float)(b1 * (double)m + (1.0 - b1) * (double)g);
    v = (float)(b2 * (double)v + (1.0 - b2) * (double)g * (double)g);
    const float denom = (float)((double)(sqrtf(v) / a.bc2_sqrt) + a.eps_d);
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(kOptBlock) void opt_adamw(OptArgs a) {
    __shared__ float lds[kOptBlock / 64];
    float coef = 1.f;
    if (a.max_norm > 0.f) {
        float s = 0.f;
        for (int i = threadIdx.x; i < a.NC; i += kOptBlock) s += a.partials[i];
        s = block_sum(s, lds);
        const float norm = sqrtf(s);
        const float c = a.max_norm / (norm + 1e-6f);
        coef = c < 1.f ? c : 1.f;
        if (!(norm == norm)) coef = norm;                       // a NaN norm poisons the update, like torch's multiply by a NaN coefficient
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out) { a.norm_out[0] = norm; a.norm_out[1] = coef; }
    }
    const int t = a.chunk_tensor[blockIdx.x];
    const int64_t off = a.chunk_off[blockIdx.x];
    const int64_t left = a.numel[t] - off;
    const int n = (int)(left < a.chunk_elems ? left : a.chunk_elems);
    float *p = reinterpret_cast<float *>(a.p[t]) + off;
    const float *g = reinterpret_cast<const float *>(a.g[t]) + off;
    float *m = reinterpret_cast<float *>(a.m[t]) + off;
    float *v = reinterpret_cast<float *>(a.v[t]) + off;
    const float step_size = (float)(a.lr_d / (double)a.bc1);
    bf16_t *sh = (a.shadow && a.shadow[t]) ? reinterpret_cast<bf16_t *>(a.shadow[t]) + off : nullptr;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)sh & 7) == 0) {
        float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        constexpr int U = 2;
        for (int i0 = threadIdx.x; i0 < (n >> 2); i0 += U * kOptBlock) {
            float4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kOptBlock < (n >> 2) ? i0 + u * kOptBlock : i0;
                pp[u] = p4[i]; gg[u] = g4[i]; mm[u] = m4[i]; vv[u] = v4[i];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kOptBlock;
                if (i >= (n >> 2)) break;
                adamw1(pp[u].x, gg[u].x * coef, mm[u].x, vv[u].x, a, step_size);
                adamw1(pp[u].y, gg[u].y * coef, mm[u].y, vv[u].y, a, step_size);
                adamw1(pp[u].z, gg[u].z * coef, mm[u].z, vv[u].z, a, step_size);
                adamw1(pp[u].w, gg[u].w * coef, mm[u].w, vv[u].w, a, step_size);
                p4[i] = pp[u]; m4[i] = mm[u]; v4[i] = vv[u];
                if (sh) reinterpret_cast<uint2 *>(sh)[i] = make_uint2(pack_bf16x2(pp[u].x, pp[u].y), pack_bf16x2(pp[u].z, pp[u].w));
            }
        }
        for (int i = (n & ~3) + threadIdx.x; i < n; i += kOptBlock) {
            float pp = p[i], mm = m[i], vv = v[i];
            adamw1(pp, g[i] * coef, mm, vv, a, step_size);
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (sh) sh[i] = (bf16_t)(pack_bf16x2(pp, 0.f) & 0xFFFFu);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += kOptBlock) {
            float pp = p[i], mm = m[i], vv = v[i];
            adamw1(pp, g[i] * coef, mm, vv, a, step_size);
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (sh) sh[i] = (bf16_t)(pack_bf16x2(pp, 0.f) & 0xFFFFu);
        }
    }
}

}  // namespace

extern "C" int mmt_clip_adamw_step(int num_chunks, int chunk_elems, const int32_t *chunk_tensor, const int64_t *chunk_offset,
                                   const int64_t *param_ptrs, const int64_t *grad_ptrs, const int64_t *exp_avg_ptrs,
                                   const int64_t *exp_avg_sq_ptrs, const int64_t *bf16_shadow_ptrs, const int64_t *numel, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int64_t step, float max_norm, float *partials, float *norm_out, void *stream) {
    if (num_chunks == 0) return MMT_OK;
    MMT_REQUIRE_PTR(chunk_tensor);
    MMT_REQUIRE_PTR(chunk_offset);
    MMT_REQUIRE_PTR(param_ptrs);
    MMT_REQUIRE_PTR(grad_ptrs);
    MMT_REQUIRE_PTR(exp_avg_ptrs);
    MMT_REQUIRE_PTR(exp_avg_sq_ptrs);
    MMT_REQUIRE_PTR(numel);
    if (num_chunks < 0 || chunk_elems <= 0 || (chunk_elems & 3) || step <= 0 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "clip_adamw_step: bad arguments (chunks=%d chunk_elems=%d step=%lld betas %g %g)", num_chunks, chunk_elems,
                         (long long)step, beta1, beta2);
    if (max_norm > 0.f) MMT_REQUIRE_PTR(partials);
    OptArgs a;
    a.p = param_ptrs; a.g = grad_ptrs; a.m = exp_avg_ptrs; a.v = exp_avg_sq_ptrs; a.shadow = bf16_shadow_ptrs; a.numel = numel;
    a.chunk_tensor = chunk_tensor; a.chunk_off = chunk_offset; a.chunk_elems = chunk_elems; a.NC = num_chunks;
    a.partials = partials; a.norm_out = norm_out;
    a.lr_d = lr; a.beta1_d = beta1; a.beta2_d = beta2; a.eps_d = eps; a.wd_d = weight_decay; a.max_norm = max_norm;
    // torch (fused_adam_utils.cuh): bias_correction1 = 1 - pow(beta1, step), bias_correction2_sqrt = sqrt(1 - pow(beta2, step)), in double then float
    a.bc1 = (float)(1.0 - pow(beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    hipStream_t st = (hipStream_t)stream;
    if (max_norm > 0.f) {
        hipLaunchKernelGGL(opt_sumsq, dim3(num_chunks), dim3(kOptBlock), 0, st, a);
        if (int rc = mmt::check_launch("clip_adamw_step(norm)")) return rc;
    }
    hipLaunchKernelGGL(opt_adamw, dim3(num_chunks), dim3(kOptBlock), 0, st, a);
    return mmt::check_launch("clip_adamw_step(update)");
}

// ---------------------------------------------------------------------------------------------------------------------
// out = in_0 + in_1 + ... + in_{n-1} (n <= 32 dense fp32 tensors of one size) in ONE pass: n reads + 1 write, where autograd's
// accumulation of n gradients is n - 1 read-read-write passes (the 24 branches of the CenterPoint head hand their input's gradient
// back separately: 22 adds of 16.8 MB per step at BASELINE configs[3]).  Summed in argument order.
namespace {

struct AddNArgs {
    const float *in[32];
    float *out;
    int64_t n4, tail0, numel;
    int n;
};

__global__ __launch_bounds__(256) void add_n_kernel(AddNArgs a) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += stride) {
        float4 s = reinterpret_cast<const float4 *>(a.in[0])[i];
        for (int k = 1; k < a.n; ++k) {
            const float4 v = reinterpret_cast<const float4 *>(a.in[k])[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4 *>(a.out)[i] = s;
    }
    if (blockIdx.x == 0)
        for (int64_t i = a.tail0 + threadIdx.x; i < a.numel; i += 256) {
            float s = a.in[0][i];
            for (int k = 1; k < a.n; ++k) s += a.in[k][i];
            a.out[i] = s;
        }
}

}  // namespace

extern "C" int mmt_add_n(int n, const void *const *inputs_host, int64_t numel, float *out, void *stream) {
    MMT_REQUIRE_PTR(inputs_host);
    MMT_REQUIRE_PTR(out);
    if (n < 1 || n > 32 || numel < 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "add_n: 1 <= n <= 32 tensors (n=%d numel=%lld)", n, (long long)numel);
    if (numel == 0) return MMT_OK;
    AddNArgs a;
    bool aligned = ((uintptr_t)out & 15) == 0;
    for (int k = 0; k < n; ++k) {
        if (inputs_host[k] == nullptr) return mmt::fail(MMT_ERR_NULL_POINTER, "add_n: input %d is NULL", k);
        a.in[k] = static_cast<const float *>(inputs_host[k]);
        aligned = aligned && ((uintptr_t)inputs_host[k] & 15) == 0;
    }
    a.out = out; a.n = n; a.numel = numel;
    a.n4 = aligned ? numel / 4 : 0;
    a.tail0 = a.n4 * 4;
    hipLaunchKernelGGL(add_n_kernel, dim3(mmt::stream_grid(a.n4 > 0 ? a.n4 : 1, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, a);
    return mmt::check_launch("add_n");
}

// A channels-last activation [rows, n * W] (n blocks of W channels per pixel) <-> n dense [rows, W] tensors, one pass either way.
// The task heads' 24 first convolutions run as ONE 64 -> 24 x 64 convolution and ONE BatchNorm (layers/heads/bev_depth_head.py); the
// 24 final convolutions each want their 64 channels as a dense tensor (split), and hand 24 separate gradients back (gather).
namespace {

struct ChanBlocksArgs {
    void *part[32];
    void *wide;
    int64_t units;      // rows * n * wb16 16-byte units in the wide tensor
    int n, wb16;        // blocks per row, 16-byte units per block row
};

template <bool SPLIT>
__global__ __launch_bounds__(256) void channel_blocks_kernel(ChanBlocksArgs a) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int per_row = a.n * a.wb16;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < a.units; u += stride) {
        const int64_t r = u / per_row;
        const int rem = (int)(u - r * per_row);
        const int k = rem / a.wb16, j = rem - k * a.wb16;
        uint4 *p = reinterpret_cast<uint4 *>(a.part[k]) + r * a.wb16 + j;
        uint4 *w = reinterpret_cast<uint4 *>(a.wide) + u;
        if (SPLIT) *p = *w;
        else *w = *p;
    }
}

int channel_blocks(const char *who, bool split, int64_t rows, int n, int block_bytes, void *wide, void *const *parts_host, void *stream) {
    if (!wide || !parts_host) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: NULL argument", who);
    if (rows < 0 || n < 1 || n > 32 || block_bytes < 16 || block_bytes % 16)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: rows >= 0, 1 <= n <= 32 blocks of a multiple of 16 bytes (rows=%lld n=%d block_bytes=%d)", who,
                         (long long)rows, n, block_bytes);
    if (rows == 0) return MMT_OK;
    ChanBlocksArgs a;
    if ((uintptr_t)wide & 15) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the wide tensor is not 16-byte aligned", who);
    for (int k = 0; k < n; ++k) {
        if (!parts_host[k]) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: part %d is NULL", who, k);
        if ((uintptr_t)parts_host[k] & 15) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: part %d is not 16-byte aligned", who, k);
        a.part[k] = parts_host[k];
    }
    a.wide = wide; a.n = n; a.wb16 = block_bytes / 16; a.units = rows * n * a.wb16;
    const dim3 grid(mmt::stream_grid(a.units, 256, 256 * 16));
    if (split) hipLaunchKernelGGL(channel_blocks_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(channel_blocks_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return mmt::check_launch(who);
}

}  // namespace

extern "C" int mmt_channel_blocks_split(int64_t rows, int n, int block_bytes, const void *wide, void *const *parts_host, void *stream) {
    return channel_blocks("channel_blocks_split", true, rows, n, block_bytes, const_cast<void *>(wide), parts_host, stream);
}

extern "C" int mmt_channel_blocks_gather(int64_t rows, int n, int block_bytes, const void *const *parts_host, void *wide, void *stream) {
    return channel_blocks("channel_blocks_gather", false, rows, n, block_bytes, wide, const_cast<void *const *>(parts_host), stream);
}
