// conv_kernels.hip -- the small kernels of the convolution plan (conv_plan.cpp): kernel
// spectrum into the FFT's k1-major order, real taps to complex, and the overlap-save store.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gcwt {

typedef float2 cf;

// k1major[k1 * 4096 + k2] = natural[k1 + P1 * k2].  grid (P / 256)
__global__ void __launch_bounds__(256) k_conv_permute(const cf* __restrict__ natural, cf* __restrict__ k1major, int p1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  k1major[i] = natural[(i >> 12) + (int64_t)p1 * (i & (kRowLenDev - 1))];
}

// dst[i] = (src[i], 0), i < m.  grid (ceil(m / 256))
__global__ void __launch_bounds__(256) k_conv_widen(const float* __restrict__ src, cf* __restrict__ dst, int64_t m) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < m) dst[i] = make_float2(src[i], 0.f);
}

// Overlap-save: sample i >= m-1 of chunk c's circular convolution is full-convolution sample
// c*step + i - (m-1); the mode's crop keeps [first, first + count).  y: [chunks here][C][P].
// grid (ceil(step / 256), C * chunks here)
__global__ void __launch_bounds__(256) k_conv_store(const cf* __restrict__ y, cf* __restrict__ out, int64_t p,
                                                    int64_t m, int64_t step, int chunk0, int n_channels,
                                                    int64_t first, int64_t count, float scale) {
  const int slot = blockIdx.y, g = slot / n_channels, ch = slot - g * n_channels;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;          // 0 .. step-1
  if (j >= step) return;
  const int64_t o = (int64_t)(chunk0 + g) * step + j - first;
  if (o < 0 || o >= count) return;
  const cf v = y[(int64_t)slot * p + (m - 1) + j];
  out[(int64_t)ch * count + o] = make_float2(v.x * scale, v.y * scale);
}

hipError_t launch_conv_permute(const cf* natural, cf* k1major, int p1, hipStream_t st) {
  hipLaunchKernelGGL(k_conv_permute, dim3((unsigned)(p1 * (kRowLenDev / 256))), dim3(256), 0, st, natural, k1major, p1);
  return hipGetLastError();
}

hipError_t launch_conv_widen(const float* src, cf* dst, int64_t m, hipStream_t st) {
  hipLaunchKernelGGL(k_conv_widen, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, src, dst, m);
  return hipGetLastError();
}

hipError_t launch_conv_store(const cf* y, cf* out, int64_t p, int64_t m, int64_t step, int chunk0,
                             int n_chunks_here, int n_channels, int64_t first, int64_t count, float scale,
                             hipStream_t st) {
  dim3 grid((unsigned)((step + 255) / 256), (unsigned)(n_channels * n_chunks_here)), block(256);
  hipLaunchKernelGGL(k_conv_store, grid, block, 0, st, y, out, p, m, step, chunk0, n_channels, first, count, scale);
  return hipGetLastError();
}

}  // namespace gcwt
