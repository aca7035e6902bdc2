// scratch micro-benchmarks (not part of the product): what bounds a 605 MB streaming write?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" {
__global__ void k_fill_gs(f4* dst, int64_t n, int nt) {
  f4 z = {1.f, 2.f, 3.f, 4.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (nt) __builtin_nontemporal_store(z, dst + i); else dst[i] = z;
  }
}
// each block writes a contiguous span of `per` float4 (tile-major like vp_bwd)
__global__ void k_fill_tiles(f4* dst, int64_t n, int nt) {
  f4 z = {1.f, 2.f, 3.f, 4.f};
  const int64_t tilev = 1024;
  int64_t ntiles = (n + tilev - 1) / tilev;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    for (int u = 0; u < 4; ++u) {
      int64_t i = t * tilev + threadIdx.x + u * 256;
      if (i < n) { if (nt) __builtin_nontemporal_store(z, dst + i); else dst[i] = z; }
    }
  }
}
// pos-dependent: reads pos (3 ints per 20 float4), stores (no gather)
__global__ void k_pos_store(f4* dst, const int* pos, int64_t n, int cv, int nt) {
  const int64_t tilev = 1024;
  int64_t ntiles = (n + tilev - 1) / tilev;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    for (int u = 0; u < 4; ++u) {
      int64_t i = t * tilev + threadIdx.x + u * 256;
      if (i < n) {
        int64_t r = i / cv;
        int b = pos[r * 3], y = pos[r * 3 + 1], x = pos[r * 3 + 2];
        f4 z = {(float)b, (float)y, (float)x, 0.f};
        if (nt) __builtin_nontemporal_store(z, dst + i); else dst[i] = z;
      }
    }
  }
}
void launch(int which, void* dst, const void* pos, int64_t n, int cv, int grid, int nt, hipStream_t st) {
  if (which == 0) hipLaunchKernelGGL(k_fill_gs, dim3(grid), dim3(256), 0, st, (f4*)dst, n, nt);
  if (which == 1) hipLaunchKernelGGL(k_fill_tiles, dim3(grid), dim3(256), 0, st, (f4*)dst, n, nt);
  if (which == 2) hipLaunchKernelGGL(k_pos_store, dim3(grid), dim3(256), 0, st, (f4*)dst, (const int*)pos, n, cv, nt);
}
}
