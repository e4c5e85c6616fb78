// detect.hip -- precision = auto: predict, per scale, what the float32 stages of the decimated path cost it, from
// the band energies of the float64 spectrum (summed where it is made: fwd64.hip, k_fwd64_rows -> row_band_sums).  The reference computes in float64
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

// The rows' band sums (kernels.h: kRowBands; written by the forward row pass, fwd64.hip: row_band_sums) added up into
// the slot's kSpecBands band energies: one thread per band, rows in row order -- the same bits every run.
__global__ void __launch_bounds__(kSpecBands) k_band_sums(const float* __restrict__ hist, int p1, int lg_p1,
                                                          float* __restrict__ bands) {
  const int b = threadIdx.x;
  const float* const hs = hist + (int64_t)blockIdx.x * p1 * kRowBands;
  float acc = 0.f;
  // bins k1 + p1 k2, k2 >= 16: band(k) = band(k2) + 16 log2 p1; the rows keep band(k2) = 64 .. 175 at entries 16 .. 127
  const int entry = b - 16 * lg_p1 - 64 + 16;
  if (entry >= 16 && entry < kRowBands)
    for (int r = 0; r < p1; ++r) acc += hs[(int64_t)r * kRowBands + entry];
  // k2 < 16: kept bin by bin (entries 0 .. 15); this band's bins below 16 p1, in order (bin 0 belongs to no band)
  float lo, hi;
  band_edges(b, &lo, &hi);
  const int k_lo = max(1, (int)ceilf(lo)), k_hi = min((int)ceilf(hi), 16 * p1);
  for (int k = k_lo; k < k_hi; ++k)
    if (spec_band((float)k) == b) acc += hs[(int64_t)(k & (p1 - 1)) * kRowBands + (k >> lg_p1)];
  bands[(int64_t)blockIdx.x * kSpecBands + b] = acc;
}

__global__ void __launch_bounds__(256) k_precision_predict(const float* __restrict__ bands, const float* __restrict__ gain,
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
  for (int i = tid; i < kSpecBands; i += 256) h[i] = bands[(int64_t)slot * kSpecBands + i];
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

hipError_t launch_band_sums(const float* hist, int p1, float* bands, int n_slots, hipStream_t st) {
  if (n_slots <= 0 || p1 <= 0) return hipSuccess;
  int lg = 0;
  while ((1 << lg) < p1) ++lg;
  if ((1 << lg) != p1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_band_sums, dim3((unsigned)n_slots), dim3(kSpecBands), 0, st, hist, p1, lg, bands);
  return hipGetLastError();
}

hipError_t launch_precision_predict(const float* bands, const float* gain, const int32_t* scale_level,
                                    const int32_t* scale_length, const PredLevel* levels, int n_scales, int n_levels, double p_true,
                                    float kappa_eps, float oob_tol, float* pred, float* dbg_level, float* dbg_scale,
                                    int n_slots, const PredSegs& segs, hipStream_t st) {
  if (n_slots <= 0 || n_scales <= 0 || n_levels <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_precision_predict, dim3(n_slots, n_levels), dim3(256), 0, st, bands, gain, scale_level, scale_length, levels, n_scales,
                     n_levels, (float)p_true, kappa_eps, oob_tol, pred, dbg_level, dbg_scale, segs);
  return hipGetLastError();
}

}  // namespace gcwt
