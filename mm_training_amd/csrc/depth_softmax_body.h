// The depth softmax's row arithmetic (csrc/depth_softmax.hip has the story): shared by the stand-alone kernels and by the
// launch that carries the plan form's calibration lookup along (csrc/lift_splat_plan.hip: lss_plan_lookup_softmax).
// Included inside each file's anonymous namespace.
#pragma once

constexpr int kGroup = 16;                 // lanes per pixel = one DPP row

struct SoftmaxArgs {
    int64_t pixels;
    int D;
    const void *logits;          // forward: [pixels] rows of D (fp32 or bf16), logit_stride elements apart
    int64_t logit_stride;
    float *probs;                // [pixels, D] fp32 (forward: written; backward: read)
    const float *oracle;         // nullable; rows oracle_stride floats apart
    int64_t oracle_stride;
    void *used;                  // forward, nullable: [pixels, D] fp32 or bf16
    const float *grad_probs;     // backward, nullable [pixels, D]
    const void *grad_used;       // backward, nullable [pixels, D] fp32 or bf16
    void *grad_logits;           // backward: [pixels, D] fp32 or bf16
};

template <bool MAX>
__device__ __forceinline__ float row16_reduce(float v) {
    // xor 1, xor 2 inside the quad, then the mirrored half row and the mirrored row: every lane of the 16 ends with the result
#define MMT_DPP_STEP(ctrl)                                                                              \
    {                                                                                                   \
        const float o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xF, 0xF, true)); \
        v = MAX ? fmaxf(v, o) : v + o;                                                                  \
    }
    MMT_DPP_STEP(0xB1)      // quad_perm [1,0,3,2]
    MMT_DPP_STEP(0x4E)      // quad_perm [2,3,0,1]
    MMT_DPP_STEP(0x141)     // row_half_mirror
    MMT_DPP_STEP(0x140)     // row_mirror
#undef MMT_DPP_STEP
    return v;
}

// element e0 .. e0 + VEC - 1 of a row -> fp32 registers; dead pieces (e0 >= D) read as `fill`
template <typename T, int VEC>
__device__ __forceinline__ void load_piece(const T *row, int e0, int D, float fill, float *dst) {
    if (e0 >= D) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = fill;
        return;
    }
    if constexpr (sizeof(T) == 4) {
        if constexpr (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(row + e0);
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        } else {
            dst[0] = row[e0];
        }
    } else {
        if constexpr (VEC == 4) {
            const uint2 t = *reinterpret_cast<const uint2 *>(row + e0);
            dst[0] = bf16_lo(t.x); dst[1] = bf16_hi(t.x); dst[2] = bf16_lo(t.y); dst[3] = bf16_hi(t.y);
        } else {
            dst[0] = __uint_as_float((unsigned)row[e0] << 16);
        }
    }
}

template <typename T, int VEC>
__device__ __forceinline__ void store_piece(T *row, int e0, int D, const float *src) {
    if (e0 >= D) return;
    if constexpr (sizeof(T) == 4) {
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(row + e0) = make_float4(src[0], src[1], src[2], src[3]);
        else row[e0] = src[0];
    } else {
        if constexpr (VEC == 4) *reinterpret_cast<uint2 *>(row + e0) = make_uint2(pack_bf16x2(src[0], src[1]), pack_bf16x2(src[2], src[3]));
        else row[e0] = (bf16_t)(pack_bf16x2(src[0], 0.f) & 0xFFFFu);
    }
}

// LT: logits type, UT: depth_used type (float / bf16_t); NV pieces of VEC elements per lane cover D <= 16 * NV * VEC.
// Workgroup `block` of `nblocks`, `nthreads` threads each: a lane group of 16 per pixel, pixels strided over all groups,
// U of a group's pixels in flight at once (their loads issued before the first is used).
template <typename LT, typename UT, int VEC, int NV, int U = 1>
__device__ __forceinline__ void softmax_fwd_rows(const SoftmaxArgs &a, int block, int nblocks, int tid, int nthreads) {
    const int grp = tid / kGroup, lane = tid % kGroup, gpb = nthreads / kGroup;
    const int D = a.D;
    const int64_t stride = (int64_t)nblocks * gpb;
    const bool has_oracle = a.oracle != nullptr;
    for (int64_t pix0 = (int64_t)block * gpb + grp; pix0 < a.pixels; pix0 += stride * U) {
        float v[U][NV * VEC], o[U][NV * VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pix = pix0 + u * stride;
            if (pix < a.pixels) {
                const LT *row = static_cast<const LT *>(a.logits) + pix * a.logit_stride;
#pragma unroll
                for (int k = 0; k < NV; ++k) load_piece<LT, VEC>(row, (k * kGroup + lane) * VEC, D, -INFINITY, v[u] + k * VEC);
                if (has_oracle) {
                    const float *orow = a.oracle + pix * a.oracle_stride;
#pragma unroll
                    for (int k = 0; k < NV; ++k) load_piece<float, VEC>(orow, (k * kGroup + lane) * VEC, D, 0.f, o[u] + k * VEC);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pix = pix0 + u * stride;
            if (pix >= a.pixels) break;           // (uniform over the 16 lanes of the group: the DPP steps below stay inside it)
            float m = v[u][0];
#pragma unroll
            for (int i = 1; i < NV * VEC; ++i) m = fmaxf(m, v[u][i]);
            m = row16_reduce<true>(m);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) {
                v[u][i] = expf(v[u][i] - m);            // dead pieces: exp(-inf) = 0
                s += v[u][i];
            }
            s = row16_reduce<false>(s);
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) v[u][i] = __fdiv_rn(v[u][i], s);
            float *prow = a.probs + pix * D;
#pragma unroll
            for (int k = 0; k < NV; ++k) store_piece<float, VEC>(prow, (k * kGroup + lane) * VEC, D, v[u] + k * VEC);
            if (a.used != nullptr) {
                if (has_oracle) {
                    float om = o[u][0];
#pragma unroll
                    for (int i = 1; i < NV * VEC; ++i) om = fmaxf(om, o[u][i]);
                    om = row16_reduce<true>(om);
                    if (om > 0.f) {                 // lss_fpn.py:429: fg_mask = torch.max(depth_oracle, dim=1).values > 0.0
#pragma unroll
                        for (int i = 0; i < NV * VEC; ++i) v[u][i] = o[u][i];
                    }
                }
                UT *urow = static_cast<UT *>(a.used) + pix * D;
#pragma unroll
                for (int k = 0; k < NV; ++k) store_piece<UT, VEC>(urow, (k * kGroup + lane) * VEC, D, v[u] + k * VEC);
            }
        }
    }
}

// ---- host side: what both entry points check
inline int softmax_common_check(const char *who, int64_t pixels, int D, int logits_dtype, int used_dtype) {
    if (pixels < 0 || D <= 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: bad sizes (pixels=%lld D=%d)", who, (long long)pixels, D);
    if (D > 512) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: D=%d exceeds 512 depth bins", who, D);
    if ((logits_dtype != MMT_DTYPE_F32 && logits_dtype != MMT_DTYPE_BF16) || (used_dtype != MMT_DTYPE_F32 && used_dtype != MMT_DTYPE_BF16))
        return mmt::fail(MMT_ERR_BAD_FLAG, "%s: dtype must be MMT_DTYPE_F32 or MMT_DTYPE_BF16", who);
    return 0;
}

// 16-byte pieces need rows that start on 16-byte (fp32) / 8-byte (bf16) boundaries and D % 4 == 0
inline bool softmax_aligned(const void *p, int64_t stride_elems, int elem_bytes) {
    return p == nullptr || ((((uintptr_t)p) % (4 * elem_bytes)) == 0 && (stride_elems % 4) == 0);
}
