// Micro-benchmark: what a no-return ds_add_f32 costs on gfx950 by address pattern (one workgroup per CU, NW waves).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_atomic.hip -o /tmp/lds_atomic ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    __shared__ float win[704 * 20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 704 * 20; i += blockDim.x) win[i] = 0.f;
    __syncthreads();
    int addr;
    if (MODE == 0) addr = wave * 64 + lane;                                  // 64 consecutive words
    else if (MODE == 1) addr = ((wave * 16 + (lane & 15)) * 17 + 4 * (lane >> 4));   // pixel on lane, stride 17, 4 channel quads
    else if (MODE == 2) addr = wave * 64;                                    // one address
    else if (MODE == 3) addr = ((wave * 16 + (lane >> 4) * 4) * 20 + (lane & 15));   // channel on lane: 4 pixels x 16 words, stride 20
    else if (MODE == 4) addr = wave * 64 + lane;                             // plain ds_write for comparison
    else addr = ((wave * 16 + (lane & 15)) * 17 + 4 * (lane >> 4));          // MODE 5: pattern 1 as read + add + write (no atomic)
    float v = (float)lane;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) win[addr + r * 0] = v + it;
        } else if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r) win[addr + r] += v;
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(win + addr + r * 20, v);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(win + addr + (MODE == 2 ? 0 : r), v);
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) out[blockIdx.x] = (float)(t1 - t0) / (float)(iters * 4);
    if (tid == 1) out[1024 + blockIdx.x] = win[lane];
}

int main() {
    float *d;
    hipMalloc(&d, 8192);
    float h[2048];
    for (int nw : {1, 4, 11, 16}) {
        for (int mode = 0; mode < 6; ++mode) {
            const int iters = 2000;
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(nw * 64), 0, 0, d, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(nw * 64), 0, 0, d, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(nw * 64), 0, 0, d, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(nw * 64), 0, 0, d, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(nw * 64), 0, 0, d, iters); break;
                default: hipLaunchKernelGGL(k<5>, dim3(256), dim3(nw * 64), 0, 0, d, iters); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(h, d, 8192, hipMemcpyDeviceToHost);
            printf("waves %2d mode %d: %.1f clock64 ticks per wave-instruction per wave (x%d waves sharing the LDS => %.1f per instr at the LDS)\n", nw, mode, h[0], nw, h[0] / nw);
        }
    }
    return 0;
}
