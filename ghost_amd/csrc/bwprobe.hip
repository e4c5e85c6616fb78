// bwprobe.hip -- measurement only (include/ghostcwt_debug.h: gcwt_debug_bandwidth): the
// HBM rates this device reaches on plain streams and on k_synth7's own store pattern,
// so that a bench line can quote its kernel against what the box in hand delivers and
// not only against the 8 TB/s data sheet.  Same kernels as tools/write_bw.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/ghostcwt_debug.h"

namespace {

__global__ void bw_fill128(float4* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  const float4 q = make_float4(v, v, v, v);
  for (; i < n; i += st) p[i] = q;
}
// four 16-byte loads in flight per thread before the first store
__global__ void bw_copy128(const float4* a, float4* b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i + 3 * st < n; i += 4 * st) {
    const float4 q0 = a[i], q1 = a[i + st], q2 = a[i + 2 * st], q3 = a[i + 3 * st];
    b[i] = q0; b[i + st] = q1; b[i + 2 * st] = q2; b[i + 3 * st] = q3;
  }
  for (; i < n; i += st) b[i] = a[i];
}
// synthesis-like pattern: a workgroup of 512 threads writes 14 runs of 32 consecutive floats
// (128 B) per 16-lane-group in each of `rows` rows that are row_len floats apart
// blockIdx.y = channel: one launch fills the chip like the synthesis does (round 1's probe
// launched 139 workgroups per channel, one channel after the other, and read 5.1 TB/s)
__global__ void bw_fill_rows(float* p, size_t row_len, int rows, float v) {
  const int lane = threadIdx.x & 31, m2 = threadIdx.x >> 5;
  const size_t col0 = (size_t)blockIdx.x * 32 * 16 * 14;
  float* const base = p + (size_t)blockIdx.y * rows * row_len + col0;
  for (int r = 0; r < rows; ++r) {
    float* q = base + (size_t)r * row_len;
    for (int m1 = 0; m1 < 14; ++m1) q[(size_t)(m2 + 16 * m1) * 32 + lane] = v;
  }
}

}  // namespace

int gcwt_internal_set_error(int code, const char* msg);

extern "C" int gcwt_debug_bandwidth(int pattern, size_t bytes, double* gb_per_s) {
  if (!gb_per_s || bytes < ((size_t)64 << 20) || pattern < 0 || pattern > 2)
    return gcwt_internal_set_error(GCWT_ERR_INVALID, "gcwt_debug_bandwidth: bad argument");
  float *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipMalloc((void**)&a, bytes);
  if (err == hipSuccess && pattern == GCWT_BW_COPY) err = hipMalloc((void**)&b, bytes);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  // best of a few grid sizes (the streaming kernels' rate depends on how many loads are in
  // flight) and of three timed passes each; the first pass of all touches the pages
  double best_ms = 1e30, moved = 0.0;
  const unsigned grids[4] = {2048, 4096, 8192, 16384};
  for (int gi = 0; gi < (pattern == GCWT_BW_SYNTH_STORES ? 1 : 4) && err == hipSuccess; ++gi) {
    for (int it = 0; it < (gi == 0 ? 4 : 3) && err == hipSuccess; ++it) {
      (void)hipEventRecord(e0, 0);
      if (pattern == GCWT_BW_FILL) {
        hipLaunchKernelGGL(bw_fill128, dim3(grids[gi]), dim3(256), 0, 0, (float4*)a, bytes / 16, 1.f);
        moved = (double)bytes;
      } else if (pattern == GCWT_BW_COPY) {
        hipLaunchKernelGGL(bw_copy128, dim3(grids[gi]), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16);
        moved = 2.0 * (double)bytes;
      } else {
        const size_t row_len = 1000000 / 32 * 32 + 32;
        const int rows = 100;
        const unsigned wgs = (unsigned)(row_len / (32 * 16 * 14));
        const size_t per_ch = (size_t)rows * row_len * 4;
        const int n_ch = (int)std::min<size_t>(120, bytes / per_ch);
        hipLaunchKernelGGL(bw_fill_rows, dim3(wgs, n_ch), dim3(512), 0, 0, a, row_len, rows, 1.f);
        moved = (double)wgs * 512 * 14 * rows * 4 * n_ch;
      }
      (void)hipEventRecord(e1, 0);
      err = hipEventSynchronize(e1);
      float ms = 0.f;
      if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
      if (gi > 0 || it > 0) best_ms = std::min(best_ms, (double)ms);
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (err != hipSuccess) return gcwt_internal_set_error(GCWT_ERR_HIP, hipGetErrorString(err));
  *gb_per_s = moved / 1e9 / (best_ms * 1e-3);
  return GCWT_OK;
}
