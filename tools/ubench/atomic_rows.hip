// Micro-benchmark: cost of flushing 320-byte fp32 rows into a [65536, 80] map with no-return global float atomics,
// for the lane layouts a lift-splat flush could use.  Build: hipcc --offload-arch=gfx950 -O3 atomic_rows.hip -o atomic_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int C = 80;

// mode 0: wave handles 4 rows as 5 full 64-lane instructions (row-contiguous dwords)
// mode 1: lanes 0..19 only, 4 instructions of 80 contiguous bytes (channel-strided registers)
// mode 2: 3 lane groups x 20 lanes, each its own row, 4 instructions (80 contiguous bytes per group)
// mode 3: lanes 0..19 only, float4-per-lane layout (4 instructions, 16-byte lane stride)
// mode 4: 3 lane groups, float4-per-lane layout
__global__ __launch_bounds__(256) void flush(int mode, int nrows, const int *cell, float *out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nw = (gridDim.x * 256) >> 6;
    if (mode == 0) {
        for (int r0 = wave * 4; r0 + 3 < nrows; r0 += nw * 4) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int e = k * 64 + lane;            // 0..319
                const int r = e / C, c = e - r * C;
                unsafeAtomicAdd(out + (size_t)cell[r0 + r] * C + c, 1.0f);
            }
        }
    } else if (mode == 1 || mode == 3) {
        for (int r = wave; r < nrows; r += nw) {
            if (lane < 20) {
                float *p = out + (size_t)cell[r] * C;
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(p + (mode == 1 ? j * 20 + lane : lane * 4 + j), 1.0f);
            }
        }
    } else {
        const int g = lane / 20, li = lane - g * 20;
        for (int r0 = wave * 3; r0 + 2 < nrows; r0 += nw * 3) {
            if (g < 3) {
                float *p = out + (size_t)cell[r0 + g] * C;
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(p + (mode == 2 ? j * 20 + li : li * 4 + j), 1.0f);
            }
        }
    }
}

// LDS float atomics: every lane group (20 lanes) adds its 4 registers to one of `nslot` 80-float rows, `iters` times;
// 256 threads = 12 lane groups.  sink keeps the compiler honest.
__global__ __launch_bounds__(256) void lds_add(int nslot, int iters, float *sink) {
    __shared__ float rows[128 * C];
    for (int i = threadIdx.x; i < 128 * C; i += 256) rows[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / 20, li = lane - g * 20;
    unsigned s = (blockIdx.x * 12 + wave * 3 + g) * 2654435761u;
    if (g < 3) {
        for (int it = 0; it < iters; ++it) {
            s = s * 1664525u + 1013904223u;
            float *p = rows + ((s >> 8) % nslot) * C + li;
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 20, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 40, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 60, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = rows[5];
}

int main(int argc, char **argv) {
    const int nrows = argc > 1 ? atoi(argv[1]) : 98304;
    const int ncell = 65536;
    std::vector<int> h(nrows);
    float *out; int *cell;
    hipMalloc(&out, (size_t)ncell * C * 4); hipMalloc(&cell, nrows * 4);
    hipMemset(out, 0, (size_t)ncell * C * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pattern = 0; pattern < 2; ++pattern) {
        srand(1);
        for (int i = 0; i < nrows; ++i) h[i] = pattern == 0 ? rand() % ncell : (i * 7 / 5) % ncell;  // random / ray-like neighbours
        hipMemcpy(cell, h.data(), nrows * 4, hipMemcpyHostToDevice);
        for (int grid : {512, 2048}) {
            for (int mode = 0; mode < 5; ++mode) {
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(flush, dim3(grid), dim3(256), 0, 0, mode, nrows, cell, out);
                hipEventRecord(e0);
                const int reps = 20;
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(flush, dim3(grid), dim3(256), 0, 0, mode, nrows, cell, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / reps;
                printf("pattern %d grid %4d mode %d: %7.2f us  %6.1f GB/s of added bytes (%d rows)\n", pattern, grid, mode, us,
                       (double)nrows * C * 4 / us * 1e-3, nrows);
            }
        }
    }
    {   // LDS atomics: 1024 workgroups (4 per CU resident, LDS 40 KB each), 200 row adds per lane group
        float *sink; hipMalloc(&sink, 4096 * 4);
        for (int nslot : {128, 16, 1}) {
            const int iters = 200, grid = 1024;
            hipLaunchKernelGGL(lds_add, dim3(grid), dim3(256), 0, 0, nslot, iters, sink);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(lds_add, dim3(grid), dim3(256), 0, 0, nslot, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / 10;
            // per CU: 4 workgroups x 4 waves x iters x 4 instructions
            printf("lds_add nslot %3d: %7.2f us for %d row adds per lane group -> %.1f clk per wave-instruction per CU (2.4 GHz), %.1f ns per row add per CU\n",
                   nslot, us, iters, us * 2400.0 / (4.0 * 4 * iters * 4), us * 1e3 / (4.0 * 12 * iters));
        }
    }
    return 0;
}
