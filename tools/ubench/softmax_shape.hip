// Micro-benchmark: the depth softmax's rows (csrc/depth_softmax_body.h) under the kernel shapes it can be launched in --
// why the launch that carries the calibration lookup (lss_plan_lookup_softmax: 1 024 threads, 128 VGPRs, 16 KB LDS, a private
// segment) takes 8.4 us for the rows the 256-thread kernel does in 6.3.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imm_training_amd/csrc tools/ubench/softmax_shape.hip mm_training_amd/csrc/mmt_common.hip -o /tmp/softmax_shape
#include "mmt_common.h"
#include <stdio.h>
#include <vector>
namespace {
#include "depth_softmax_body.h"

// MODE bits: 1 = hold 128 VGPRs, 2 = 16 KB of static LDS, 4 = a private segment
template <int THREADS, int U, int MODE>
__global__ __launch_bounds__(THREADS) void k(SoftmaxArgs a, int never) {
    if (MODE & 2) {
        __shared__ float pad[4096];
        if (never) pad[threadIdx.x] = 1.f;
        if (never == 2) a.probs[0] = pad[threadIdx.x ^ 1];
    }
    if (MODE & 4) {
        if (never) { volatile float arr[80]; for (int i = 0; i < 80; ++i) arr[i] = i; a.probs[1] = arr[never & 63]; }
    }
    if (MODE & 1) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    softmax_fwd_rows<float, float, 4, 2, U>(a, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, THREADS);
}
}  // namespace

template <typename K>
static float run(K kern, dim3 grid, dim3 block, SoftmaxArgs a, const char *what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < 60; ++i) {
        hipExtLaunchKernelGGL(kern, grid, block, 0, 0, e0, e1, 0, a, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 10) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-64s grid %4u x %4u: median %.2f us\n", what, grid.x, block.x, ts[ts.size() / 2] * 1e3f);
    return ts[ts.size() / 2];
}

int main() {
    const int64_t pixels = 4 * 6 * 16 * 44; const int D = 112;
    float *logits, *probs, *oracle, *used;
    hipMalloc(&logits, pixels * D * 4); hipMalloc(&probs, pixels * D * 4); hipMalloc(&oracle, pixels * D * 4); hipMalloc(&used, pixels * D * 4);
    hipMemset(logits, 0, pixels * D * 4); hipMemset(oracle, 0, pixels * D * 4);
    SoftmaxArgs a{pixels, D, logits, D, probs, oracle, D, used, nullptr, nullptr, nullptr};
    const unsigned g256 = (unsigned)((pixels * 16 + 255) / 256), g1024 = (unsigned)((pixels * 16 + 1023) / 1024);
    run(k<256, 1, 0>, dim3(g256), dim3(256), a, "256 threads, the stand-alone kernel's shape");
    run(k<1024, 1, 0>, dim3(g1024), dim3(1024), a, "1024 threads, a row per group");
    run(k<1024, 2, 0>, dim3(251), dim3(1024), a, "1024 threads, two rows in flight, 251 workgroups");
    run(k<1024, 2, 1>, dim3(251), dim3(1024), a, "  + 128 VGPRs");
    run(k<1024, 2, 2>, dim3(251), dim3(1024), a, "  + 16 KB LDS");
    run(k<1024, 2, 4>, dim3(251), dim3(1024), a, "  + private segment");
    run(k<1024, 2, 7>, dim3(251), dim3(1024), a, "  + all three (the rider's shape)");
    run(k<1024, 1, 7>, dim3(g1024), dim3(1024), a, "  all three, a row per group, 264 workgroups");
    run(k<512, 2, 7>, dim3(502), dim3(512), a, "512 threads, all three, 502 workgroups");
    run(k<256, 1, 7>, dim3(g256), dim3(256), a, "256 threads, all three");
    run(k<256, 1, 1>, dim3(g256), dim3(256), a, "256 threads + 128 VGPRs");
    run(k<256, 1, 4>, dim3(g256), dim3(256), a, "256 threads + private segment");
    return 0;
}
