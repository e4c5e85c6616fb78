// Write-only and copy bandwidth of the device, the ceilings k_synth7 (51 GB of stores per
// launch) is judged against.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/write_bw.hip -o /tmp/write_bw && /tmp/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill32(float* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = v;
}
__global__ void fill128(float4* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  const float4 q = make_float4(v, v, v, v);
  for (; i < n; i += st) p[i] = q;
}
// synthesis-like pattern: a workgroup of 512 threads writes 14 rows of 32 consecutive
// floats (128 B) in each of `rows` scale rows that are row_len floats apart
__global__ void fill_rows(float* p, size_t row_len, int rows, float v) {
  const int lane = threadIdx.x & 31, m2 = threadIdx.x >> 5;       // 16 row groups
  const size_t col0 = (size_t)blockIdx.x * 32 * 16 * 14;          // this workgroup's samples
  for (int r = 0; r < rows; ++r) {
    float* q = p + (size_t)r * row_len + col0;
    for (int m1 = 0; m1 < 14; ++m1) q[(size_t)(m2 + 16 * m1) * 32 + lane] = v;
  }
}
__global__ void copy128(const float4* a, float4* b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = a[i];
}
__global__ void read128(const float4* a, float* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i < n; i += st) { const float4 q = a[i]; acc += q.x + q.y + q.z + q.w; }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)48 << 30;   // 48 GiB
  float *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes / 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double gb, auto&& launch) {
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-34s %8.3f ms  %7.1f GB/s\n", name, best, gb / (best * 1e-3)); fflush(stdout);
  };
  const double gb = bytes / 1e9;
  timeit("hipMemsetAsync 48 GiB", gb, [&] { CK(hipMemsetAsync(a, 0, bytes, 0)); });
  timeit("fill b32  (2048 x 256 grid-stride)", gb, [&] { hipLaunchKernelGGL(fill32, dim3(2048), dim3(256), 0, 0, a, bytes / 4, 1.f); });
  timeit("fill b32  (one float per thread)", gb / 4, [&] { hipLaunchKernelGGL(fill32, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, a, bytes / 16, 1.f); });
  timeit("fill b128 (2048 x 256 grid-stride)", gb, [&] { hipLaunchKernelGGL(fill128, dim3(2048), dim3(256), 0, 0, (float4*)a, bytes / 16, 1.f); });
  {
    const size_t row_len = 1000000 / 32 * 32 + 32;       // ~1e6-sample rows
    const int rows = 100;
    const unsigned wgs = (unsigned)(row_len / (32 * 16 * 14));
    const double g = (double)wgs * 512 * 14 * rows * 4 * 120 / 1e9;
    timeit("rows pattern (120 ch x 100 rows)", g, [&] {
      for (int c = 0; c < 120; ++c)
        hipLaunchKernelGGL(fill_rows, dim3(wgs), dim3(512), 0, 0, a + (size_t)c * rows * row_len, row_len, rows, 1.f);
    });
  }
  timeit("copy b128 12 GiB -> 12 GiB (r+w)", 2 * gb / 4, [&] { hipLaunchKernelGGL(copy128, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 64); });
  timeit("read b128 48 GiB", gb, [&] { hipLaunchKernelGGL(read128, dim3(2048), dim3(256), 0, 0, (const float4*)a, b, bytes / 16); });
  return 0;
}
