// Producers of voxel_pooling's operands: quantise (lss_fpn.py:461-462), fused
// frustum geometry (lss_fpn.py:328-361) and the lift + channels-last layout step
// (lss_fpn.py:441-463).  All HBM-streaming kernels; fp32 throughout.
#include "mmt_camera.h"

namespace {

constexpr int kBlock = 256;

typedef mmt::CamGrid GridQ;   // lo = fp32(voxel_coord - fp32(voxel_size/2)), vs (mmt_camera.h)

__device__ __forceinline__ int quantize_one(float v, float lo, float vs) { return mmt_quantize_exact(v, lo, vs); }

__global__ __launch_bounds__(kBlock) void quantize_kernel(int64_t n, const float *xyz, GridQ q,
                                                          int32_t *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += stride) {
        const float x = xyz[t * 3], y = xyz[t * 3 + 1], z = xyz[t * 3 + 2];
        out[t * 3] = quantize_one(x, q.lo[0], q.vs[0]);
        out[t * 3 + 1] = quantize_one(y, q.lo[1], q.vs[1]);
        out[t * 3 + 2] = quantize_one(z, q.lo[2], q.vs[2]);
    }
}

// One workgroup column per camera (blockIdx.y): the 3x4 matrix is wave-uniform.
__global__ __launch_bounds__(kBlock) void frustum_geometry_kernel(int64_t S, const float4 *frustum,
                                                                  const float *combine, GridQ q,
                                                                  int32_t *geom, float *xyz_out) {
    const int bn = blockIdx.y;
    const float *M = combine + (int64_t)bn * 16;
    float m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = M[i];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < S; s += stride) {
        const float4 f = frustum[s];
        // p = (u*d, v*d, d, w), un-contracted k-ordered dot products (mmt_camera.h), then the exact quantise
        float r[3];
        mmt_cam_xyz(m, f.x, f.y, f.z, f.w, r);
        const int64_t t = (int64_t)bn * S + s;
        geom[t * 3] = quantize_one(r[0], q.lo[0], q.vs[0]);
        geom[t * 3 + 1] = quantize_one(r[1], q.lo[1], q.vs[1]);
        geom[t * 3 + 2] = quantize_one(r[2], q.lo[2], q.vs[2]);
        if (xyz_out) {
            xyz_out[t * 3] = r[0];
            xyz_out[t * 3 + 1] = r[1];
            xyz_out[t * 3 + 2] = r[2];
        }
    }
}

// ---------------------------------------------------------------------------
// Lift: feats[bn,d,s,c] = depth[bn,d,s] * context[bn,c,s].
// Workgroup = (camera bn, tile of kTile positions, range of kDRange depth bins).
// The context tile is transposed once through LDS ([c][s] -> [s][c], rows padded to
// C+4 floats so they stay 16-byte aligned); every depth bin then emits one fully
// contiguous kTile*C*4-byte block of the channels-last output with 16-byte
// non-temporal stores.
constexpr int kTile = 64;
constexpr int kDRange = 4;

template <bool VEC4, int C4T>
__global__ __launch_bounds__(kBlock) void lift_kernel(int D, int HW, int C, const float *depth,
                                                      const float *context, float *feats) {
    extern __shared__ __align__(16) float lds[];
    const int CP = C + 4;
    float *ctx = lds;                 // [kTile][CP]
    float *dep = lds + kTile * CP;    // [kDRange][kTile]
    const int bn = blockIdx.z;
    const int s0 = blockIdx.x * kTile;
    const int d0 = blockIdx.y * kDRange;
    const int ns = (HW - s0) < kTile ? (HW - s0) : kTile;
    const int nd = (D - d0) < kDRange ? (D - d0) : kDRange;
    const int tid = threadIdx.x;

    for (int i = tid; i < C * kTile; i += kBlock) {
        const int c = i / kTile, j = i - c * kTile;
        if (j < ns) ctx[j * CP + c] = context[((int64_t)bn * C + c) * HW + s0 + j];
    }
    for (int i = tid; i < nd * kTile; i += kBlock) {
        const int dd = i / kTile, j = i - dd * kTile;
        if (j < ns) dep[i] = depth[((int64_t)bn * D + d0 + dd) * HW + s0 + j];
    }
    __syncthreads();

    if (VEC4) {
        const int C4 = C4T > 0 ? C4T : C >> 2;   // compile-time for the common widths: no runtime division
        const int nvec = ns * C4;
        for (int dd = 0; dd < nd; ++dd) {
            float4 *dst = reinterpret_cast<float4 *>(feats + (((int64_t)bn * D + d0 + dd) * HW + s0) * C);
            for (int i = tid; i < nvec; i += kBlock) {
                const int j = i / C4, c4 = i - j * C4;
                const float dv = dep[dd * kTile + j];
                const float4 cv = *reinterpret_cast<const float4 *>(ctx + j * CP + c4 * 4);
                mmt_nt_store4(make_float4(dv * cv.x, dv * cv.y, dv * cv.z, dv * cv.w), dst + i);
            }
        }
    } else {
        const int nel = ns * C;
        for (int dd = 0; dd < nd; ++dd) {
            float *dst = feats + (((int64_t)bn * D + d0 + dd) * HW + s0) * C;
            for (int i = tid; i < nel; i += kBlock) {
                const int j = i / C, c = i - j * C;
                dst[i] = dep[dd * kTile + j] * ctx[j * CP + c];
            }
        }
    }
}

// bf16 storage of the lifted features: same tiling, the product is rounded once to bf16 (nearest even) and a lane
// emits 8 channels = one 16-byte vector.  C % 8 == 0.
template <int C8T>
__global__ __launch_bounds__(kBlock) void lift_kernel_bf16(int D, int HW, int C, const float *depth,
                                                           const float *context, bf16_t *feats) {
    extern __shared__ __align__(16) float lds[];
    const int CP = C + 4;
    float *ctx = lds;                 // [kTile][CP]
    float *dep = lds + kTile * CP;    // [kDRange][kTile]
    const int bn = blockIdx.z;
    const int s0 = blockIdx.x * kTile;
    const int d0 = blockIdx.y * kDRange;
    const int ns = (HW - s0) < kTile ? (HW - s0) : kTile;
    const int nd = (D - d0) < kDRange ? (D - d0) : kDRange;
    const int tid = threadIdx.x;
    for (int i = tid; i < C * kTile; i += kBlock) {
        const int c = i / kTile, j = i - c * kTile;
        if (j < ns) ctx[j * CP + c] = context[((int64_t)bn * C + c) * HW + s0 + j];
    }
    for (int i = tid; i < nd * kTile; i += kBlock) {
        const int dd = i / kTile, j = i - dd * kTile;
        if (j < ns) dep[i] = depth[((int64_t)bn * D + d0 + dd) * HW + s0 + j];
    }
    __syncthreads();
    const int C8 = C8T > 0 ? C8T : C >> 3;
    const int nvec = ns * C8;
    for (int dd = 0; dd < nd; ++dd) {
        mmt_u32x4 *dst = reinterpret_cast<mmt_u32x4 *>(feats + (((int64_t)bn * D + d0 + dd) * HW + s0) * C);
        for (int i = tid; i < nvec; i += kBlock) {
            const int j = i / C8, c8 = i - j * C8;
            const float dv = dep[dd * kTile + j];
            const float4 a = *reinterpret_cast<const float4 *>(ctx + j * CP + c8 * 8);
            const float4 b = *reinterpret_cast<const float4 *>(ctx + j * CP + c8 * 8 + 4);
            mmt_u32x4 r;
            r.x = pack_bf16x2(dv * a.x, dv * a.y); r.y = pack_bf16x2(dv * a.z, dv * a.w);
            r.z = pack_bf16x2(dv * b.x, dv * b.y); r.w = pack_bf16x2(dv * b.z, dv * b.w);
            __builtin_nontemporal_store(r, dst + i);
        }
    }
}

// Lift backward.  grad_depth[bn,d,s] = sum_c g[bn,d,s,c]*ctx[bn,c,s];
//                 grad_context[bn,c,s] = sum_d g[bn,d,s,c]*depth[bn,d,s].
// Workgroup = (camera, tile of NG consecutive positions) and walks ALL depth bins, so
// grad_context needs no cross-workgroup sum.  Like the pooling kernels a wave is split
// into G = 64/(C/4) lane groups of C/4 lanes, one group per position, one float4 column
// per lane: per depth bin a group reads one contiguous C*4-byte row of g (the NG groups
// together NG*C*4 contiguous bytes), keeps its grad_context float4 in registers over the
// whole depth loop (4 rows in flight), and reduces the row's dot product with ctx across
// its lanes by two DPP quad adds + one LDS float add per quad.
// GT = float, or bf16_t for a bf16-stored gradient (a lane then reads its 4 channels as one 8-byte vector).
template <typename GT, int C4T>
__global__ __launch_bounds__(kBlock) void lift_backward_vec4(int D, int HW, int C, const float *depth,
                                                             const float *context, const GT *g,
                                                             float *grad_depth, float *grad_context) {
    extern __shared__ __align__(16) float lds[];
    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4;
    const int NG = (kBlock / 64) * G;            // positions per workgroup
    float *dep = lds;                            // [D][NG] depth tile
    float *gd = dep + D * NG;                    // [D][NG] grad_depth accumulators
    const int bn = blockIdx.y;
    const int s0 = blockIdx.x * NG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / C4, li = lane - grp * C4;
    const int j = wave * G + grp;                // position inside the tile
    const bool active = grp < G && (s0 + j) < HW;

    for (int i = tid; i < D * NG; i += kBlock) {
        const int d = i / NG, jj = i - d * NG;
        dep[i] = (s0 + jj) < HW ? depth[((int64_t)bn * D + d) * HW + s0 + jj] : 0.f;
        gd[i] = 0.f;
    }
    float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const float *cp = context + ((int64_t)bn * C + li * 4) * HW + s0 + j;
        cx = make_float4(cp[0], cp[HW], cp[2 * (int64_t)HW], cp[3 * (int64_t)HW]);
    }
    __syncthreads();

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const GT *src = g + (((int64_t)bn * D) * HW + s0 + j) * C + li * 4;
    const int64_t dstride = (int64_t)HW * C;
    for (int d0 = 0; d0 < D; d0 += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (active && d0 + u < D) {
                if constexpr (sizeof(GT) == 2) {
                    const uint2 r = *reinterpret_cast<const uint2 *>(src + (d0 + u) * dstride);
                    v[u] = make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y));
                } else {
                    v[u] = *reinterpret_cast<const float4 *>(src + (d0 + u) * dstride);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = d0 + u;
            if (d < D) {
                const float dv = dep[d * NG + (active ? j : 0)];
                acc.x += v[u].x * dv; acc.y += v[u].y * dv; acc.z += v[u].z * dv; acc.w += v[u].w * dv;
                float dot = v[u].x * cx.x + v[u].y * cx.y + v[u].z * cx.z + v[u].w * cx.w;
                dot += __shfl_xor(dot, 1);       // quad reduction (lane groups start on multiples of 4
                dot += __shfl_xor(dot, 2);       // when C/4 is a multiple of 4; otherwise see the host check)
                if (active && (li & 3) == 0) atomicAdd(&gd[d * NG + j], dot);
            }
        }
    }
    if (active) {
        float *gp = grad_context + ((int64_t)bn * C + li * 4) * HW + s0 + j;
        gp[0] = acc.x; gp[HW] = acc.y; gp[2 * (int64_t)HW] = acc.z; gp[3 * (int64_t)HW] = acc.w;
    }
    __syncthreads();
    for (int i = tid; i < D * NG; i += kBlock) {
        const int d = i / NG, jj = i - d * NG;
        if ((s0 + jj) < HW) grad_depth[((int64_t)bn * D + d) * HW + s0 + jj] = gd[i];
    }
}

// any C (slow path): one element per lane, LDS float atomics for the dot product
constexpr int kBTile = 16;

__global__ __launch_bounds__(kBlock) void lift_backward_kernel(int D, int HW, int C,
                                                               const float *depth,
                                                               const float *context,
                                                               const float *g, float *grad_depth,
                                                               float *grad_context) {
    extern __shared__ __align__(16) float lds[];
    const int CP = C + 4;
    float *ctx = lds;                    // [kBTile][CP]
    float *gctx = ctx + kBTile * CP;     // [kBTile][CP]  accumulated grad_context
    float *gd = gctx + kBTile * CP;      // [2][kBTile]   grad_depth of the bin in flight
    const int bn = blockIdx.y;
    const int s0 = blockIdx.x * kBTile;
    const int ns = (HW - s0) < kBTile ? (HW - s0) : kBTile;
    const int tid = threadIdx.x;
    for (int i = tid; i < C * kBTile; i += kBlock) {
        const int c = i / kBTile, j = i - c * kBTile;
        if (j < ns) ctx[j * CP + c] = context[((int64_t)bn * C + c) * HW + s0 + j];
    }
    for (int i = tid; i < kBTile * CP; i += kBlock) gctx[i] = 0.f;
    if (tid < 2 * kBTile) gd[tid] = 0.f;
    __syncthreads();
    const int nel = ns * C;
    for (int d = 0; d < D; ++d) {
        float *gdb = gd + (d & 1) * kBTile;
        const float *src = g + (((int64_t)bn * D + d) * HW + s0) * C;
        const float *drow = depth + ((int64_t)bn * D + d) * HW + s0;
        for (int i = tid; i < nel; i += kBlock) {
            const int j = i / C, c = i - j * C;
            const float gv = src[i];
            atomicAdd(&gdb[j], gv * ctx[j * CP + c]);
            gctx[j * CP + c] += gv * drow[j];  // (j,c) is owned by exactly one lane
        }
        __syncthreads();
        if (tid < ns) {
            grad_depth[((int64_t)bn * D + d) * HW + s0 + tid] = gdb[tid];
            gdb[tid] = 0.f;
        }
    }
    __syncthreads();
    for (int i = tid; i < C * kBTile; i += kBlock) {
        const int c = i / kBTile, j = i - c * kBTile;
        if (j < ns) grad_context[((int64_t)bn * C + c) * HW + s0 + j] = gctx[j * CP + c];
    }
}

int make_grid(const float *vc, const float *vs, GridQ *q) {
    mmt::make_cam_grid(vc, vs, q);
    return 0;
}

}  // namespace

extern "C" int mmt_quantize_geometry(int64_t n, const float *xyz, const float *vc_host,
                                     const float *vs_host, int32_t *geom, void *stream) {
    MMT_REQUIRE_PTR(xyz);
    MMT_REQUIRE_PTR(vc_host);
    MMT_REQUIRE_PTR(vs_host);
    MMT_REQUIRE_PTR(geom);
    if (n < 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "quantize_geometry: negative point count");
    if (n == 0) return MMT_OK;
    GridQ q;
    make_grid(vc_host, vs_host, &q);
    // (a four-points-per-thread form with 16-byte accesses was measured: 11.6 us against 8.6 us for this one at 1.9 M points)
    hipLaunchKernelGGL(quantize_kernel, dim3(mmt::stream_grid(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, n, xyz, q, geom);
    return mmt::check_launch("quantize_geometry");
}

extern "C" int mmt_frustum_geometry(int BN, int64_t S, const float *frustum, const float *combine,
                                    const float *vc_host, const float *vs_host, int32_t *geom,
                                    float *xyz_out, void *stream) {
    MMT_REQUIRE_PTR(frustum);
    MMT_REQUIRE_PTR(combine);
    MMT_REQUIRE_PTR(vc_host);
    MMT_REQUIRE_PTR(vs_host);
    MMT_REQUIRE_PTR(geom);
    if (BN <= 0 || S <= 0 || BN > 65535) return mmt::fail(MMT_ERR_BAD_SHAPE, "frustum_geometry: bad sizes BN=%d S=%lld", BN, (long long)S);
    if (((uintptr_t)frustum & 15) != 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "frustum_geometry: frustum must be 16-byte aligned");
    GridQ q;
    make_grid(vc_host, vs_host, &q);
    dim3 grid((unsigned)mmt::stream_grid(S, kBlock, 1024), (unsigned)BN);
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (micro-benchmarks only)
    seq.launch(true, frustum_geometry_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, S,
               reinterpret_cast<const float4 *>(frustum), combine, q, geom, xyz_out);
    return mmt::check_launch("frustum_geometry");
}

extern "C" int mmt_lift_features(int BN, int D, int HW, int C, const float *depth,
                                 const float *context, float *feats, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(feats);
    if (BN <= 0 || D <= 0 || HW <= 0 || C <= 0 || BN > 65535)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "lift_features: bad sizes");
    const size_t lds = ((size_t)kTile * (C + 4) + (size_t)kDRange * kTile) * 4;
    if (lds > 160 * 1024) return mmt::fail(MMT_ERR_TOO_LARGE, "lift_features: C=%d too large for the LDS tile", C);
    dim3 grid((unsigned)mmt::ceil_div(HW, kTile), (unsigned)mmt::ceil_div(D, kDRange), (unsigned)BN);
    const bool vec4 = (C % 4 == 0) && (((uintptr_t)feats & 15) == 0);
    if (vec4 && C == 80) hipLaunchKernelGGL((lift_kernel<true, 20>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, feats);
    else if (vec4 && C == 64) hipLaunchKernelGGL((lift_kernel<true, 16>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, feats);
    else if (vec4) hipLaunchKernelGGL((lift_kernel<true, 0>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, feats);
    else hipLaunchKernelGGL((lift_kernel<false, 0>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, feats);
    return mmt::check_launch("lift_features");
}

extern "C" int mmt_lift_features_backward(int BN, int D, int HW, int C, const float *depth,
                                          const float *context, const float *grad_feats,
                                          float *grad_depth, float *grad_context, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_feats);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    if (BN <= 0 || D <= 0 || HW <= 0 || C <= 0 || BN > 65535)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "lift_features_backward: bad sizes");
    // vector path: rows are whole float4 columns and every lane group starts on a quad boundary
    const int C4 = C / 4;
    const bool vec4 = (C % 16 == 0) && C <= 256 && (((uintptr_t)grad_feats & 15) == 0);
    if (vec4) {
        const int NG = (kBlock / 64) * (64 / C4);
        const size_t lds = (size_t)2 * D * NG * 4;
        if (lds <= 64 * 1024) {
            dim3 grid((unsigned)mmt::ceil_div(HW, NG), (unsigned)BN);
            if (C == 80) hipLaunchKernelGGL((lift_backward_vec4<float, 20>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, grad_feats, grad_depth, grad_context);
            else if (C == 64) hipLaunchKernelGGL((lift_backward_vec4<float, 16>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, grad_feats, grad_depth, grad_context);
            else hipLaunchKernelGGL((lift_backward_vec4<float, 0>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, grad_feats, grad_depth, grad_context);
            return mmt::check_launch("lift_features_backward(vec4)");
        }
    }
    const size_t lds = ((size_t)2 * kBTile * (C + 4) + 2 * kBTile) * 4;
    if (lds > 160 * 1024) return mmt::fail(MMT_ERR_TOO_LARGE, "lift_features_backward: C too large");
    dim3 grid((unsigned)mmt::ceil_div(HW, kBTile), (unsigned)BN);
    hipLaunchKernelGGL(lift_backward_kernel, grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C,
                       depth, context, grad_feats, grad_depth, grad_context);
    return mmt::check_launch("lift_features_backward");
}

// ---- bf16 storage of the lifted feature matrix: depth / context and their gradients stay fp32 (they come
// from and go back to the fp32 dense nets), feats [BN, D, HW, C] and its gradient are bf16.  C % 16 == 0, C <= 256.
extern "C" int mmt_lift_features_bf16(int BN, int D, int HW, int C, const float *depth,
                                      const float *context, uint16_t *feats, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(feats);
    if (BN <= 0 || D <= 0 || HW <= 0 || C <= 0 || BN > 65535)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "lift_features_bf16: bad sizes");
    if (C % 8 != 0 || (((uintptr_t)feats & 15) != 0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "lift_features_bf16: needs C %% 8 == 0 and 16-byte aligned feats");
    const size_t lds = ((size_t)kTile * (C + 4) + (size_t)kDRange * kTile) * 4;
    if (lds > 160 * 1024) return mmt::fail(MMT_ERR_TOO_LARGE, "lift_features_bf16: C=%d too large for the LDS tile", C);
    dim3 grid((unsigned)mmt::ceil_div(HW, kTile), (unsigned)mmt::ceil_div(D, kDRange), (unsigned)BN);
    if (C == 80) hipLaunchKernelGGL((lift_kernel_bf16<10>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, feats);
    else if (C == 64) hipLaunchKernelGGL((lift_kernel_bf16<8>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, feats);
    else hipLaunchKernelGGL((lift_kernel_bf16<0>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, feats);
    return mmt::check_launch("lift_features_bf16");
}

extern "C" int mmt_lift_features_backward_bf16(int BN, int D, int HW, int C, const float *depth,
                                               const float *context, const uint16_t *grad_feats,
                                               float *grad_depth, float *grad_context, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_feats);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    if (BN <= 0 || D <= 0 || HW <= 0 || C <= 0 || BN > 65535)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "lift_features_backward_bf16: bad sizes");
    if (C % 16 != 0 || C > 256 || (((uintptr_t)grad_feats & 7) != 0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "lift_features_backward_bf16: needs C %% 16 == 0, C <= 256, 8-byte aligned grad_feats");
    const int C4 = C / 4;
    const int NG = (kBlock / 64) * (64 / C4);
    const size_t lds = (size_t)2 * D * NG * 4;
    if (lds > 64 * 1024) return mmt::fail(MMT_ERR_TOO_LARGE, "lift_features_backward_bf16: D too large for the LDS tile");
    dim3 grid((unsigned)mmt::ceil_div(HW, NG), (unsigned)BN);
    if (C == 80) hipLaunchKernelGGL((lift_backward_vec4<bf16_t, 20>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, grad_feats, grad_depth, grad_context);
    else if (C == 64) hipLaunchKernelGGL((lift_backward_vec4<bf16_t, 16>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, grad_feats, grad_depth, grad_context);
    else hipLaunchKernelGGL((lift_backward_vec4<bf16_t, 0>), grid, dim3(kBlock), lds, (hipStream_t)stream, D, HW, C, depth, context, grad_feats, grad_depth, grad_context);
    return mmt::check_launch("lift_features_backward_bf16");
}
