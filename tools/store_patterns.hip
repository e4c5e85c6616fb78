// What costs k_synth7's store pattern 25 % against a linear fill?  Variants of the pattern
// (dword stores, 256 B per wave-instruction, 28 KB tiles, 100 rows 4 MB apart):
//   hipcc -O3 --offload-arch=gfx950 tools/store_patterns.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill32_linear(float* p, size_t n, float v) {      // dword stores, grid-stride
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = v;
}
// rows pattern; row_order 0: every workgroup walks rows 0..rows-1; 1: starts at its own row
__global__ void fill_rows(float* p, size_t row_len, int rows, float v, int row_order, int per_row_reps) {
  const int lane = threadIdx.x & 31, m2 = threadIdx.x >> 5;
  const size_t col0 = (size_t)blockIdx.x * 32 * 16 * 14;
  for (int rr = 0; rr < rows; ++rr) {
    const int r = row_order ? (rr + blockIdx.x) % rows : rr;
    float* q = p + (size_t)r * row_len + col0;
    for (int m1 = 0; m1 < 14; ++m1) q[(size_t)(m2 + 16 * m1) * 32 + lane] = v;
  }
}
__global__ void fill_rows_y(float* p, size_t row_len, int rows, float v) {   // blockIdx.y = channel
  const int lane = threadIdx.x & 31, m2 = threadIdx.x >> 5;
  const size_t col0 = (size_t)blockIdx.x * 32 * 16 * 14;
  float* base = p + (size_t)blockIdx.y * rows * row_len + col0;
  for (int r = 0; r < rows; ++r) {
    float* q = base + (size_t)r * row_len;
    for (int m1 = 0; m1 < 14; ++m1) q[(size_t)(m2 + 16 * m1) * 32 + lane] = v;
  }
}
// the same bytes, but a thread stores 4 consecutive floats (dwordx4): 1 KB per wave-instruction
__global__ void fill_rows_x4(float* p, size_t row_len, int rows, float v) {
  const size_t col0 = (size_t)blockIdx.x * 32 * 16 * 14;
  const float4 q4 = make_float4(v, v, v, v);
  for (int r = 0; r < rows; ++r) {
    float4* q = reinterpret_cast<float4*>(p + (size_t)r * row_len + col0);
    for (int i = threadIdx.x; i < 32 * 16 * 14 / 4; i += 512) q[i] = q4;
  }
}

int main() {
  const size_t bytes = (size_t)48 << 30;
  float* a; CK(hipMalloc(&a, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double gb, auto&& launch) {
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
    }
    printf("%-64s %8.3f ms  %7.1f GB/s\n", name, best, gb / (best * 1e-3)); fflush(stdout);
  };
  const double gb = bytes / 1e9;
  timeit("linear fill, dword stores, 8192 x 256", gb, [&] { hipLaunchKernelGGL(fill32_linear, dim3(8192), dim3(256), 0, 0, a, bytes / 4, 1.f); });
  const size_t row_len = 1000000 / 32 * 32 + 32;
  const unsigned wgs = (unsigned)(row_len / (32 * 16 * 14));
  for (int rows : {1, 10, 100}) {
    for (int order = 0; order < 2; ++order) {
      if (rows == 1 && order) continue;
      const int n_ch = 120 * 100 / rows;
      const double g = (double)wgs * 512 * 14 * rows * 4 * n_ch / 1e9;
      char name[128]; snprintf(name, sizeof name, "rows pattern, %3d rows per launch x %5d launches, order %d", rows, n_ch, order);
      timeit(name, g, [&] {
        for (int c = 0; c < n_ch; ++c)
          hipLaunchKernelGGL(fill_rows, dim3(wgs), dim3(512), 0, 0, a + (size_t)c * rows * row_len, row_len, rows, 1.f, order, 1);
      });
    }
  }
  {
    const double g = (double)wgs * 512 * 14 * 100 * 4 * 120 / 1e9;
    timeit("rows pattern, 100 rows, dwordx4 stores", g, [&] {
      for (int c = 0; c < 120; ++c) hipLaunchKernelGGL(fill_rows_x4, dim3(wgs), dim3(512), 0, 0, a + (size_t)c * 100 * row_len, row_len, 100, 1.f);
    });
    // one launch for all channels (grid.y): the chip is full, no launch gaps
    timeit("rows pattern, 100 rows, ONE launch (139 x 120 workgroups)", g, [&] {
      hipLaunchKernelGGL(fill_rows_y, dim3(wgs, 120), dim3(512), 0, 0, a, row_len, 100, 1.f);
    });
  }
  return 0;
}
