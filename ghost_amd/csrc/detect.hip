// detect.hip -- precision = auto: predict, per scale, what the float32 stages of the decimated path cost it, from
// the band energies of the float64 spectrum (fwd64.hip: k_fwd64_rows, hist).  The reference computes in float64
// (transforms.py:142-143, convolution.py:68-77) and does not care how far a band lies below the rest of the
// recording; the float32 level transform and block spectra do: their rounding is white at ~2^-24 of everything
// the level's x_R contains, so a scale whose own output is D times weaker than its level's content loses
// ~1.6e-7 D of its peak (profiles/r04_dynamic_range.md).  Scales predicted above the threshold are made again
// from float64 spectra by the exact paths (api.cpp: reroute); the others keep the fast path.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gcwt {

namespace {

__device__ __forceinline__ void band_edges(int b, float* lo, float* hi) {
  const int e = b >> 4, m = b & 15;
  const float base = __builtin_ldexpf(1.0f, e);
  *lo = base * (1.0f + (float)m * 0.0625f);
  *hi = base * (1.0f + (float)(m + 1) * 0.0625f);
}

// fraction of [lo, hi) inside [a, b)
__device__ __forceinline__ float cover(float lo, float hi, float a, float b) {
  const float x = fminf(hi, b) - fmaxf(lo, a);
  return x <= 0.f ? 0.f : x / (hi - lo);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

}  // namespace

// Band energies of the spectrum (kernels.h: spec_band): one workgroup per (row k1, slot) reads the row's 2048 bins
// k1 + p1 k2, k2 < 2048 -- the positive half; consecutive k2 are p1 bins apart, so a thread's eight consecutive k2
// lie in one band as a rule -- and leaves the row's kSpecBands sums: 0.54 GB read per headline step.
__global__ void __launch_bounds__(256) k_spectrum_bands(const float2* __restrict__ x, int64_t x_cstride, int p1,
                                                        float* __restrict__ hist) {
  __shared__ float hl[kSpecBands];
  const int tid = threadIdx.x, row = blockIdx.x;
  for (int i = tid; i < kSpecBands; i += 256) hl[i] = 0.f;
  const float4* src = reinterpret_cast<const float4*>(x + (int64_t)blockIdx.y * x_cstride + (int64_t)row * kRowLenDev) + 4 * tid;
  const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];     // bins k2 = 8 tid .. 8 tid + 7
  __syncthreads();
  const float e[8] = {q0.x * q0.x + q0.y * q0.y, q0.z * q0.z + q0.w * q0.w, q1.x * q1.x + q1.y * q1.y, q1.z * q1.z + q1.w * q1.w,
                      q2.x * q2.x + q2.y * q2.y, q2.z * q2.z + q2.w * q2.w, q3.x * q3.x + q3.y * q3.y, q3.z * q3.z + q3.w * q3.w};
  const int64_t k0 = (int64_t)row + (int64_t)p1 * (8 * tid);
  const int ba = k0 > 0 ? spec_band((float)k0) : -1, bz = spec_band((float)(k0 + 7 * (int64_t)p1));
  if (ba == bz) {
    atomicAdd(hl + ba, ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7])));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t k = k0 + (int64_t)p1 * j;
      if (k > 0) atomicAdd(hl + spec_band((float)k), e[j]);
    }
  }
  __syncthreads();
  float* const hg = hist + ((int64_t)blockIdx.y * gridDim.x + row) * kSpecBands;
  for (int i = tid; i < kSpecBands; i += 256) hg[i] = hl[i];
}

__global__ void __launch_bounds__(256) k_precision_predict(const float* __restrict__ hist, int n_rows, const float* __restrict__ gain,
                                                           const int32_t* __restrict__ scale_level,
                                                           const int32_t* __restrict__ scale_length,
                                                           const PredLevel* __restrict__ levels, int n_scales,
                                                           int n_levels, float p_true, float kappa_eps, float oob_tol,
                                                           float* __restrict__ pred, float* __restrict__ dbg_level,
                                                           float* __restrict__ dbg_scale, const PredSegs segs) {
  __shared__ float h[kSpecBands];
  __shared__ float hout[kSpecBands];   // what the current level leaves out of band b: (1 - weight) h
  __shared__ float dens[256];
  __shared__ float gmin[kSpecBands];
  __shared__ int band_of[256];
  __shared__ float red[4];
  const int tid = threadIdx.x, slot = blockIdx.x;
  for (int i = tid; i < kSpecBands; i += 256) {                  // the rows' band sums, added in row order
    const float* hr = hist + (int64_t)slot * n_rows * kSpecBands + i;
    float acc = 0.f;
    for (int r = 0; r < n_rows; ++r) acc += hr[(int64_t)r * kSpecBands];
    h[i] = acc;
  }
  __syncthreads();
  {
    const int l = blockIdx.y;                                    // one workgroup per (slot, level)
    const PredLevel lv = levels[l];
    if (lv.decimation <= 0) return;                              // no spectral scale uses this level
    const float top = p_true / (float)lv.decimation;            // x_R holds the bins [-U, top - U)
    const float per_bin = top * (1.0f / 256.0f);                // spectrum bins per bin of the level's 256-point grid
    const float U = (float)lv.band_shift * per_bin;
    // what the level's float32 stages see
    float e = 0.f;
    for (int b = tid; b < kSpecBands; b += 256) {
      float lo, hi;
      band_edges(b, &lo, &hi);
      float w = cover(lo, hi, 0.f, top - U) + (U > 0.f ? cover(lo, hi, 0.f, U) : 0.f);
      if (lv.k1 > lv.k0) {                                       // the level's low cut (kernels.hip: row_taper)
        const float c = 0.5f * (lo + hi);
        const float t = c <= lv.k0 ? 0.f : (c >= lv.k1 ? 1.f : 0.5f - 0.5f * __cosf(3.14159265f * (c - lv.k0) / (lv.k1 - lv.k0)));
        w *= t * t;
      }
      e += w * h[b];
      hout[b] = (1.0f - fminf(w, 1.0f)) * h[b];
    }
    const float e_level = block_sum(e, red);
    if (dbg_level && tid == 0) dbg_level[(int64_t)slot * n_levels + l] = e_level;
    // energy per bin of the level's grid, from the band its centre lies in (negative frequencies mirror)
    {
      const float kc = fabsf(((float)(tid - lv.band_shift) + 0.5f) * per_bin);
      const int b = spec_band(fmaxf(kc, 1.0f));
      float lo, hi;
      band_edges(b, &lo, &hi);
      dens[tid] = h[b] / fmaxf(1.0f, hi - lo) * per_bin;
      band_of[tid] = b;
    }
    __syncthreads();
    for (int s = 0; s < n_scales; ++s) {
      if (scale_level[s] != l) continue;                         // workgroup-uniform
      const float g = gain[(int64_t)s * 256 + tid];
      const float g2 = g * g;
      // the scale's own energy, conservatively: a band's content counts with the SMALLEST gain the scale has on the
      // band's bins (a line may sit anywhere in its band, and a steep skirt -- gamma = 6 -- drops tenfold across one)
      gmin[band_of[tid]] = 3.0e38f;
      __syncthreads();
      atomicMin(reinterpret_cast<unsigned*>(gmin) + band_of[tid], __float_as_uint(g2));
      __syncthreads();
      const float e_s = block_sum(gmin[band_of[tid]] * dens[tid], red);
      const float w_s = block_sum(g2, red) * (1.0f / 256.0f);
      // what the level leaves out, as the reference's L-tap kernel answers to it: flat side lobes above the band;
      // below it the response of a zero-mean kernel rises linearly from zero frequency to its first side lobe at
      // half a bin of the L-point grid, k = P / (2 L)
      float eo = 0.f;
      {
        const float rise = 2.0f * (float)scale_length[s] / p_true;
        for (int b = tid; b < kSpecBands; b += 256) {
          float lo, hi;
          band_edges(b, &lo, &hi);
          const float c = 0.5f * (lo + hi);
          const float t = (c < top - U && lv.band_shift == 0) ? fminf(1.0f, c * rise) : 1.0f;
          eo += t * t * hout[b];
        }
      }
      const float e_out = block_sum(eo, red);
      if (tid == 0) {
        // (1) the float32 rounding of the level's stages, white at kappa_eps of everything its x_R holds, through
        // the scale's noise bandwidth; (2) what the level leaves out -- below its low cut, above its band -- reaches
        // the reference's result through the side lobes of its L-tap kernel (transforms.py:187-204: below oob_tol of
        // the peak there) and never reaches this path's
        const float p1 = e_s > 0.f ? kappa_eps * sqrtf(e_level * w_s / e_s) : 0.f;
        const float p2 = e_s > 0.f ? oob_tol * sqrtf(e_out / e_s) : 0.f;
        const float p = fmaxf(p1, p2);
        if (dbg_scale) { dbg_scale[((int64_t)slot * 2) * n_scales + s] = p1; dbg_scale[((int64_t)slot * 2 + 1) * n_scales + s] = p2; }
        // pred[segment][channel][scale]: one writer each (a scale belongs to one level)
        const int g = slot / segs.n_channels, ch = slot - g * segs.n_channels;
        pred[((int64_t)segs.seg[g] * segs.n_channels + ch) * n_scales + s] = p;
      }
    }
    __syncthreads();
  }
}

hipError_t launch_spectrum_bands(const float2* x, int64_t x_cstride, int p1, float* hist, int n_slots, hipStream_t st) {
  if (n_slots <= 0 || p1 <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_spectrum_bands, dim3((unsigned)p1, (unsigned)n_slots), dim3(256), 0, st, x, x_cstride, p1, hist);
  return hipGetLastError();
}

hipError_t launch_precision_predict(const float* hist, int n_rows, const float* gain, const int32_t* scale_level,
                                    const int32_t* scale_length, const PredLevel* levels, int n_scales, int n_levels, double p_true,
                                    float kappa_eps, float oob_tol, float* pred, float* dbg_level, float* dbg_scale,
                                    int n_slots, const PredSegs& segs, hipStream_t st) {
  if (n_slots <= 0 || n_scales <= 0 || n_levels <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_precision_predict, dim3(n_slots, n_levels), dim3(256), 0, st, hist, n_rows, gain, scale_level, scale_length, levels, n_scales,
                     n_levels, (float)p_true, kappa_eps, oob_tol, pred, dbg_level, dbg_scale, segs);
  return hipGetLastError();
}

}  // namespace gcwt
