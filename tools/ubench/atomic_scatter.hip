// Micro-benchmark: chip-wide rate of SCATTERED device-scope atomics, one per lane, every lane of a wave in a different
// cache line -- what bounds vox_link (csrc/lidar_voxelize.hip: one 64-bit atomicExch per LiDAR point on a dense per-cell
// table).  Cases: returning 64-bit exchange (what vox_link issues), returning 32-bit exchange, no-return 32-bit add;
// table sizes 8 MB (4 x 512 x 512 cells x 8 B: the cfg4 table) and 64 MB; 160 k (cfg4: 4 x 40 k points) and 1.28 M operations.
// Build: hipcc --offload-arch=gfx950 -O3 atomic_scatter.hip -o atomic_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void xchg64(int n, const unsigned *idx, unsigned long long *tab, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += atomicExch(&tab[idx[i]], (unsigned long long)i);
    if (acc == 0xdeadbeefcafeull) *sink = acc;
}
__global__ __launch_bounds__(256) void xchg32(int n, const unsigned *idx, unsigned *tab, unsigned *sink) {
    unsigned acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += atomicExch(&tab[idx[i]], (unsigned)i);
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ __launch_bounds__(256) void add32(int n, const unsigned *idx, unsigned *tab) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&tab[idx[i]], 1u);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const size_t table_bytes[2] = {8u << 20, 64u << 20};
    const int ops[2] = {160000, 1280000};
    void *tab, *sink;
    unsigned *idx;
    CK(hipMalloc(&tab, table_bytes[1]));
    CK(hipMalloc(&sink, 16));
    CK(hipMalloc(&idx, sizeof(unsigned) * ops[1]));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int ti = 0; ti < 2; ++ti) {
        for (int oi = 0; oi < 2; ++oi) {
            const int n = ops[oi];
            for (int kind = 0; kind < 3; ++kind) {
                const size_t entries = table_bytes[ti] / (kind == 0 ? 8 : 4);
                std::vector<unsigned> h(n);
                unsigned s = 12345u + 977u * (ti * 4 + oi);
                for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned)(((unsigned long long)(s >> 4) * entries) >> 28); }
                CK(hipMemcpy(idx, h.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice));
                CK(hipMemset(tab, 0, table_bytes[ti]));
                const int grid = (n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096;
                float best = 1e9f, sum = 0.f;
                const int reps = 20;
                for (int r = 0; r < reps + 3; ++r) {
                    CK(hipEventRecord(e0));
                    if (kind == 0) hipLaunchKernelGGL(xchg64, dim3(grid), dim3(256), 0, 0, n, idx, (unsigned long long *)tab, (unsigned long long *)sink);
                    else if (kind == 1) hipLaunchKernelGGL(xchg32, dim3(grid), dim3(256), 0, 0, n, idx, (unsigned *)tab, (unsigned *)sink);
                    else hipLaunchKernelGGL(add32, dim3(grid), dim3(256), 0, 0, n, idx, (unsigned *)tab);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 3) { sum += ms; best = ms < best ? ms : best; }
                }
                const char *names[3] = {"atomicExch u64 (returning)", "atomicExch u32 (returning)", "atomicAdd u32 (no return)"};
                printf("%-28s table %3zu MB  %8d ops: mean %7.2f us  best %7.2f us  -> %6.2f G atomics/s (best)\n", names[kind], table_bytes[ti] >> 20, n,
                       sum / reps * 1e3f, best * 1e3f, n / (best * 1e-3f) / 1e9f);
            }
        }
    }
    return 0;
}
