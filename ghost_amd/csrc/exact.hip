// exact.hip -- the full-band path: scales whose L-tap kernel (as the reference truncates
// it) answers at every frequency, so that no decimated band holds it.  Per scale and
// segment: W = IFFT_P(X * H)[0:N_e] with H the exact response of the kernel
// (morse_exact.h) on the whole P-point grid, negative frequencies included -- the
// reference's fastconv of that scale (convolution.py:68-87) as one circular convolution.
//   k_fullband_filter   H[k] / P on the k1-major grid of the two-pass FFT
//   k_fullband_mul      Z = X * H for every workspace slot
//   (inverse FFT: the row and column passes of kernels.hip)
//   k_fullband_store    crop to the segment's window, |.| / |.|^2 / complex, store
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "morse_exact.h"

namespace gcwt {

typedef float2 cf;

// element i = k1 * 4096 + k2 of the k1-major layout is bin k = k1 + P1 k2.  grid (P / 256)
__global__ void __launch_bounds__(256) k_fullband_filter(cf* __restrict__ h, const BankScale* __restrict__ sc,
                                                         int scale, const double* __restrict__ amps,
                                                         int p1) {
  const BankScale p = sc[scale];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k = (i >> 12) + (int64_t)p1 * (i & (kRowLenDev - 1));
  const int64_t P = (int64_t)p1 * kRowLenDev;
  const double g = exact_gain(amps + p.amp_offset, p.bin_lo, p.n_bins, p.length, k, P) / (double)P;
  double sn, cs;
  sincospi(-2.0 * (double)k / (double)P * p.half_delay, &sn, &cs);
  h[i] = make_float2((float)(g * cs), (float)(g * sn));
}

// grid (P / 512, slots): two complex elements per thread
__global__ void __launch_bounds__(256) k_fullband_mul(const float4* __restrict__ x, const float4* __restrict__ h,
                                                      float4* __restrict__ z, int64_t half_p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float4 a = x[(int64_t)blockIdx.y * half_p + i], b = h[i];
  z[(int64_t)blockIdx.y * half_p + i] =
      make_float4(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x, a.z * b.z - a.w * b.w, a.z * b.w + a.w * b.z);
}

// y: [slots][P] natural order; segment-local samples [w_lo, w_hi) go to the scale's row.
// grid (ceil(longest window / 256), slots)
template <int MODE>
__global__ void __launch_bounds__(256) k_fullband_store(const cf* __restrict__ y, float* __restrict__ out,
                                                        int64_t p, int scale, int n_scales,
                                                        int64_t row_len, const SegOut seg) {
  constexpr int kElem = MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1;
  const int c = blockIdx.y;
  const int g = c / seg.n_channels, ch = c - g * seg.n_channels;
  const int64_t n = seg.w_lo[g] + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= seg.w_hi[g]) return;
  const cf v = y[(int64_t)c * p + n];
  float* o = out + (((int64_t)ch * n_scales + scale) * row_len + seg.seg_col[g] + n) * kElem;
  if (MODE == GCWT_OUT_AMPLITUDE_F32) o[0] = sqrtf(v.x * v.x + v.y * v.y);
  else if (MODE == GCWT_OUT_POWER_F32) o[0] = v.x * v.x + v.y * v.y;
  else { o[0] = v.x; o[1] = v.y; }
}

hipError_t launch_fullband_filter(cf* h, const BankScale* sc, int scale, const double* amps, int p1,
                                  hipStream_t st) {
  hipLaunchKernelGGL(k_fullband_filter, dim3((unsigned)(p1 * (kRowLenDev / 256))), dim3(256), 0, st, h,
                     sc, scale, amps, p1);
  return hipGetLastError();
}

hipError_t launch_fullband_mul(const cf* x, const cf* h, cf* z, int64_t p, int n_slots, hipStream_t st) {
  hipLaunchKernelGGL(k_fullband_mul, dim3((unsigned)(p / 512), n_slots), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(h),
                     reinterpret_cast<float4*>(z), p / 2);
  return hipGetLastError();
}

hipError_t launch_fullband_store(int mode, const cf* y, float* out, int64_t p, int scale, int n_scales,
                                 int64_t row_len, const SegOut& seg, int n_segments, hipStream_t st) {
  int64_t longest = 0;
  for (int g = 0; g < n_segments; ++g) longest = std::max(longest, seg.w_hi[g] - seg.w_lo[g]);
  if (longest <= 0) return hipSuccess;
  dim3 grid((unsigned)((longest + 255) / 256), seg.n_channels * n_segments), block(256);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_fullband_store<GCWT_OUT_AMPLITUDE_F32>), grid, block, 0, st, y, out, p, scale,
                       n_scales, row_len, seg);
  else if (mode == GCWT_OUT_POWER_F32)
    hipLaunchKernelGGL((k_fullband_store<GCWT_OUT_POWER_F32>), grid, block, 0, st, y, out, p, scale,
                       n_scales, row_len, seg);
  else
    hipLaunchKernelGGL((k_fullband_store<GCWT_OUT_COMPLEX_C64>), grid, block, 0, st, y, out, p, scale,
                       n_scales, row_len, seg);
  return hipGetLastError();
}

}  // namespace gcwt
