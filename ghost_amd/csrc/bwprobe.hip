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

// k_synthi's store pattern (profiles/r04_store_study.md, tools/store_geom.hip): 256-thread workgroups, three per CU
// (42 KB of LDS each), a workgroup owns `visit` samples of a channel's rows and walks the level's 15 rows four at a
// time; inside a (row, range) visit its four waves write 1 KB runs round-robin with 16-byte nt stores.  Items fastest
// over the grid, as the kernel launches at 128 channels.  grid (ceil(n / visit), channels)
__global__ void __launch_bounds__(256, 3) bw_fill_synthi(float* __restrict__ out, int64_t pitch, int n_samples, int visit,
                                                         int n_rows, int rows_total) {
  extern __shared__ char lds_[];
  (void)lds_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t col0 = (int64_t)blockIdx.x * visit;
  const int len = min(visit, n_samples - (int)col0);
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 val = {1.f, 2.f, 3.f, 4.f};
  for (int row = 0; row < n_rows; ++row) {
    float* dst = out + ((int64_t)blockIdx.y * rows_total + row) * pitch + col0;
    for (int wt = wave; wt * 256 < len; wt += 4) {
      const int smp = wt * 256 + 4 * lane;
      if (smp + 4 <= len) __builtin_nontemporal_store(val, reinterpret_cast<v4*>(dst + smp));
    }
  }
}

// Whole-result properties (bench.py, after the timed steps): values that are not finite, and values of
// channel c >= distinct that differ in any bit from the same row of channel c % distinct (the bench tiles
// `distinct` recordings over its channels: equal inputs must give equal rows whatever workgroup made them).
// grid (chunks, rows), rows = n_channels * rows_per_channel
__global__ void __launch_bounds__(256) k_check_rows(const uint32_t* __restrict__ out, int64_t pitch, int64_t n_valid,
                                                    int rows_per_channel, int distinct,
                                                    unsigned long long* __restrict__ counts) {
  const int64_t row = blockIdx.y;
  const int c = (int)(row / rows_per_channel), s = (int)(row - (int64_t)c * rows_per_channel);
  const uint32_t* p = out + row * pitch;
  const uint32_t* twin = c >= distinct ? out + ((int64_t)(c % distinct) * rows_per_channel + s) * pitch : nullptr;
  unsigned bad = 0, diff = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_valid; i += (int64_t)gridDim.x * 256) {
    const uint32_t v = p[i];
    bad += (v & 0x7f800000u) == 0x7f800000u;          // Inf or NaN
    if (twin) diff += v != twin[i];
  }
  for (int off = 32; off > 0; off >>= 1) { bad += __shfl_down(bad, off, 64); diff += __shfl_down(diff, off, 64); }
  if ((threadIdx.x & 63) == 0) {
    if (bad) atomicAdd(&counts[0], (unsigned long long)bad);
    if (diff) atomicAdd(&counts[1], (unsigned long long)diff);
  }
}

}  // namespace

int gcwt_internal_set_error(int code, const char* msg);

extern "C" int gcwt_debug_check_output(const void* out_device, int64_t row_pitch_floats, int64_t n_valid_floats,
                                       int32_t rows_per_channel, int32_t n_channels, int32_t distinct,
                                       int64_t* n_nonfinite, int64_t* n_mismatched) {
  if (!out_device || !n_nonfinite || !n_mismatched || row_pitch_floats < n_valid_floats || n_valid_floats <= 0 ||
      rows_per_channel <= 0 || n_channels <= 0 || distinct <= 0 ||
      (int64_t)rows_per_channel * n_channels > 65535)
    return gcwt_internal_set_error(GCWT_ERR_INVALID, "gcwt_debug_check_output: bad argument");
  unsigned long long* d = nullptr;
  hipError_t err = hipMalloc((void**)&d, 16);
  if (err == hipSuccess) err = hipMemset(d, 0, 16);
  unsigned long long h[2] = {0, 0};
  if (err == hipSuccess) {
    const unsigned chunks = (unsigned)std::min<int64_t>(64, (n_valid_floats + 4095) / 4096);
    const int64_t rows = (int64_t)rows_per_channel * n_channels;     // grid.y: at most 65535 rows per call
    hipLaunchKernelGGL(k_check_rows, dim3(chunks, (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0, 0,
                       (const uint32_t*)out_device, row_pitch_floats, n_valid_floats, rows_per_channel, distinct, d);
    err = hipGetLastError();
    if (err == hipSuccess) err = hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (err != hipSuccess) return gcwt_internal_set_error(GCWT_ERR_HIP, hipGetErrorString(err));
  *n_nonfinite = (int64_t)h[0];
  *n_mismatched = (int64_t)h[1];
  return GCWT_OK;
}

extern "C" int gcwt_debug_bandwidth(int pattern, size_t bytes, double* gb_per_s) {
  if (!gb_per_s || bytes < ((size_t)64 << 20) || pattern < 0 || pattern > 3)
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
  if (err == hipSuccess && pattern == GCWT_BW_SYNTHI_STORES)
    err = hipFuncSetAttribute((const void*)bw_fill_synthi, hipFuncAttributeMaxDynamicSharedMemorySize, 42 * 1024);
  for (int gi = 0; gi < (pattern >= GCWT_BW_SYNTH_STORES ? 1 : 4) && err == hipSuccess; ++gi) {
    for (int it = 0; it < (gi == 0 ? 4 : 3) && err == hipSuccess; ++it) {
      (void)hipEventRecord(e0, 0);
      if (pattern == GCWT_BW_FILL) {
        hipLaunchKernelGGL(bw_fill128, dim3(grids[gi]), dim3(256), 0, 0, (float4*)a, bytes / 16, 1.f);
        moved = (double)bytes;
      } else if (pattern == GCWT_BW_COPY) {
        hipLaunchKernelGGL(bw_copy128, dim3(grids[gi]), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16);
        moved = 2.0 * (double)bytes;
      } else if (pattern == GCWT_BW_SYNTHI_STORES) {
        // the R = 32 level of the headline: 15 rows of a channel's 100, 53 KB visits (two blocks of 212 x 32 samples)
        const int64_t pitch = 1000032;
        const int n = 1000000, visit = 13568, n_rows = 15, rows_total = 100;
        const int n_ch = (int)std::min<size_t>(128, bytes / ((size_t)rows_total * pitch * 4));
        hipLaunchKernelGGL(bw_fill_synthi, dim3((n + visit - 1) / visit, n_ch), dim3(256), 42 * 1024, 0, a, pitch, n, visit,
                           n_rows, rows_total);
        moved = (double)n_ch * n_rows * n * 4.0;
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
