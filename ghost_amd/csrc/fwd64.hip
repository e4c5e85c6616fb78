// fwd64.hip -- the forward FFT of a recording in float64 (round 4, precision = high).
//
// Why: the reference works in float64 (transforms.py:142-143, convolution.py:68-77).  Every
// float32 stage of a transform carries the recording's WHOLE dynamic range: its rounding noise is
// white at ~1e-7 of the total, which a 1/f^3 background, mains interference 100 x the signal or
// an electrode offset put above 1e-5 of a quiet band (profiles/r04_spectrum_classes_before.json).
// The forward transform is the one stage that cannot be protected by a filter in front of it, so
// it runs in float64: x - mean in float64, both passes in float64 with a float64 intermediate,
// and only the finished spectrum is rounded to float32 -- per BIN, i.e. relative to that
// frequency's own content.  Downstream, every level low-cuts its slice of the spectrum below the
// band of its scales (api.cpp: level taper) before its float32 inverse transform.
//
// Same two-pass layout as the float32 path (kernels.hip): P = P1 * 4096,
//   pass A  Y[k1][n2] = W_P^(-n2 k1) sum_n1 x[4096 n1 + n2] W_P1^(-n1 k1)      (columns, float64 out)
//   pass B  X~[k1][k2] = sum_n2 Y[k1][n2] W_4096^(-n2 k2) = X[k1 + P1 k2]       (rows, float32 out)
// with the same register radix-16 FFT256 (16 threads x 16 points, one LDS exchange), the same
// real-input packing (two real columns, or two real subsequences of a column, per complex
// transform) and the same reflected rows for the upper half of a real signal's spectrum.
// Twiddles come from two float64 tables (W_4096^a and W_2^24^b: any W_P^m is one product).
#include <hip/hip_runtime.h>

#include <cmath>

#include "kernels.h"

namespace gcwt {

typedef double2 cd;

__device__ __forceinline__ cd dmul(cd a, cd b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cd dadd(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd dsub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cd dmul_mi(cd a) { return make_double2(a.y, -a.x); }   // times -i
__device__ __forceinline__ cd dconj(cd a) { return make_double2(a.x, -a.y); }

// forward DFT4 / DFT16 (exp(-2 pi i n k / N)), natural order in and out: kernels.hip dft4 / dft16
__device__ __forceinline__ void d_dft4(cd& a, cd& b, cd& c, cd& d) {
  const cd s0 = dadd(a, c), s1 = dsub(a, c), s2 = dadd(b, d), s3 = dmul_mi(dsub(b, d));
  a = dadd(s0, s2);
  c = dsub(s0, s2);
  b = dadd(s1, s3);
  d = dsub(s1, s3);
}

__device__ __forceinline__ void d_dft16(cd v[16]) {
  const double c1 = 0.92387953251128673848, s1 = 0.38268343236508977173;   // cos, sin(pi/8)
  const double h = 0.70710678118654752440;
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) d_dft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
  v[5] = dmul(v[5], make_double2(c1, -s1));
  v[9] = dmul(v[9], make_double2(h, -h));
  v[13] = dmul(v[13], make_double2(s1, -c1));
  v[6] = dmul(v[6], make_double2(h, -h));
  v[10] = dmul_mi(v[10]);
  v[14] = dmul(v[14], make_double2(-h, -h));
  v[7] = dmul(v[7], make_double2(s1, -c1));
  v[11] = dmul(v[11], make_double2(-h, -h));
  v[15] = dmul(v[15], make_double2(-c1, s1));
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) d_dft4(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
  cd t;
#define GCWT_SWAP(a, b) t = v[a]; v[a] = v[b]; v[b] = t;
  GCWT_SWAP(1, 4) GCWT_SWAP(2, 8) GCWT_SWAP(3, 12) GCWT_SWAP(6, 9) GCWT_SWAP(7, 13) GCWT_SWAP(11, 14)
#undef GCWT_SWAP
}

__device__ __forceinline__ void d_dft2(cd& a, cd& b) {
  const cd s = dadd(a, b), d = dsub(a, b);
  a = s;
  b = d;
}

template <int Q>
__device__ __forceinline__ void d_dft_small(cd v[Q]) {
  if (Q == 2) d_dft2(v[0], v[1]);
  else if (Q == 4) d_dft4(v[0], v[1], v[2], v[3]);
  else if (Q == 16) d_dft16(v);
}

// exchange planes of the FFT256: 17 doubles per row of 16, columns 273 doubles apart (lanes run
// over the columns)
constexpr int kDPitch = 17;
constexpr int kDCol = 273;   // = 1 (mod 16): the 16 lanes of a 64-bit LDS access (one t, columns s) fall on 16 different bank pairs
constexpr size_t kDPlaneBytes = (size_t)16 * kDCol * sizeof(double);      // one plane of 16 columns
constexpr size_t kFwd64Lds = 2 * kDPlaneBytes + 256 * sizeof(cd);         // planes + twiddle table

// 256-point forward FFT by 16 threads: v[j] = in[t + 16 j] -> out[t + 16 j]; tw_t[16 m2] = W256^(-t m2)
// from an LDS table.  One __syncthreads inside.
__device__ __forceinline__ void d_fft256(cd v[16], const cd* tw_t, double* ex_re, double* ex_im, int t) {
  d_dft16(v);
#pragma unroll
  for (int m2 = 0; m2 < 16; ++m2) {
    const cd u = dmul(v[m2], tw_t[16 * m2]);
    ex_re[t * kDPitch + m2] = u.x;
    ex_im[t * kDPitch + m2] = u.y;
  }
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1)
    v[k1] = make_double2(ex_re[k1 * kDPitch + t], ex_im[k1 * kDPitch + t]);
  d_dft16(v);
}

// exp(-2 pi i num / 2^lg), lg <= 24: tw_hi[a] = exp(-2 pi i a / 4096), tw_lo[b] = exp(-2 pi i b / 2^24)
__device__ __forceinline__ cd d_phase(const cd* __restrict__ tw_hi, const cd* __restrict__ tw_lo, int64_t num, int lg) {
  const uint32_t m = (uint32_t)(num & (((int64_t)1 << lg) - 1)) << (24 - lg);
  return dmul(tw_hi[m >> 12], tw_lo[m & 4095]);
}

// weight of segment sample n: 1, except on the faded edges of a time block (kernels.h: SegIn::ramp_lo / ramp_hi;
// planner.h: EpochPlan::ramp_*): the C2 "smootherstep" 10 t^3 - 15 t^4 + 6 t^5 over the ramp.  The ramps are a per cent
// of a block: callers test `seg_in_ramp` for the whole wave first (one vote) and only then pay for the weights --
// evaluated for every sample the polynomial made the 2^22-point pass 42 % slower.
struct SegRamp {
  int lo, hi_start, n_valid;      // samples [0, lo) and [hi_start, n_valid) are faded
  double inv_lo, inv_hi;
};
__device__ __forceinline__ SegRamp seg_ramp(const SegIn& segs, int g) {
  SegRamp r;
  r.lo = (int)segs.ramp_lo[g];
  r.n_valid = (int)segs.n_valid[g];
  r.hi_start = r.n_valid - (int)segs.ramp_hi[g];
  r.inv_lo = r.lo > 0 ? 1.0 / (double)r.lo : 0.0;
  r.inv_hi = segs.ramp_hi[g] > 0 ? 1.0 / (double)segs.ramp_hi[g] : 0.0;
  return r;
}
__device__ __forceinline__ bool seg_in_ramp(const SegRamp& r, int n) { return n < r.lo || n >= r.hi_start; }
__device__ __forceinline__ double seg_weight(const SegRamp& r, int n) {
  double t;
  if (n < r.lo) t = (double)n * r.inv_lo;
  else if (n >= r.hi_start) t = (double)(r.n_valid - n) * r.inv_hi;
  else return 1.0;
  t = fmax(t, 0.0);
  return t * t * t * (10.0 + t * (6.0 * t - 15.0));
}

__device__ __forceinline__ void d_fill_twl(cd* twl, const cd* __restrict__ tw_hi, int tid) {
  twl[tid] = tw_hi[((((tid & 15) * (tid >> 4)) & 255)) << 4];            // W256^(-t j) at [j][t]
}

// ---------------------------------------------------------------------------
// Pass A, len = 256 (P = 2^20): two real columns per FFT256 (kernels.hip: k_fft_cols256_real2).
// Rows 0 .. 128 are written; with rows_out = 256 the mirrored rows too (plans with full-band
// scales read the whole spectrum).  grid (ld / 32, slots), dynamic LDS kFwd64Lds
// ---------------------------------------------------------------------------
// fold_mean: the workgroup's share of the channel's sum into its slot of the partial sums (kernels.h: kSumParts), in
// a fixed order -- lanes by butterfly, the four waves in turn.  scratch: four doubles of LDS nobody else is using.
__device__ __forceinline__ void fold_store(double acc, double* scratch, double* __restrict__ fold_parts, int n_channels,
                                           int ch, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((tid & 63) == 0) scratch[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0)
    fold_parts[(int64_t)n_channels + (int64_t)ch * kSumParts + blockIdx.x] = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

constexpr int kColsTiles = 4;     // column tiles (32 real columns each) a workgroup walks: the next tile's samples
                                  // are in flight while this one is transformed (2 workgroups = 8 waves per CU are
                                  // all the LDS allows: without the prefetch the waves wait 71 % of their cycles)
__global__ void __launch_bounds__(256) k_fwd64_cols256_real2(const float* __restrict__ in, cd* __restrict__ out,
                                                             int ld, int64_t in_cstride, int64_t out_cstride,
                                                             int lg_p, const cd* __restrict__ tw_hi,
                                                             const cd* __restrict__ tw_lo,
                                                             const double* __restrict__ sums, double inv_n,
                                                             const SegIn segs, int rows_out, double* __restrict__ fold_parts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* const ex_re = reinterpret_cast<double*>(smem);
  double* const ex_im = ex_re + 16 * kDCol;
  cd* const tile = reinterpret_cast<cd*>(smem);                          // aliases the planes
  cd* const twl = reinterpret_cast<cd*>(smem + 2 * kDPlaneBytes);
  const int c = blockIdx.y, tid = threadIdx.x;                           // c: workspace slot
  const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
  // (segment-local sample indices are below P <= 2^24: 32-bit arithmetic, a third of the kernel's instructions
  // were 64-bit clamps and compares)
  const int n_valid = (int)segs.n_valid[g], n_lead = (int)segs.n_lead[g];
  const SegRamp ramp = seg_ramp(segs, g);
  const float* x = in + (int64_t)ch * in_cstride + segs.x_off[g];
  // transforms.py:142-143: float64 copy minus the global mean -- or, fold_parts (kernel-uniform; kernels.h:
  // launch_fwd64_cols): the samples as they are, their sum left for the row pass to take the mean's transform out
  const double mean = fold_parts ? 0.0 : sums[ch] * inv_n;
  double acc = 0.0;
  const int s = tid & 15, t = tid >> 4;
  d_fill_twl(twl, tw_hi, tid);
  const int top = n_valid > 0 ? n_valid - 1 : 0;
  const int tile0 = blockIdx.x * kColsTiles, n_tiles = min(kColsTiles, ld / 32 - tile0);
  float ra[16], rb[16];
  auto fetch = [&](int col0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = (t + 16 * j) * ld + col0 + 2 * s;
      ra[j] = x[min(max(n, n_lead), top)];           // clamped: no branch around the loads
      rb[j] = x[min(max(n + 1, n_lead), top)];
    }
  };
  fetch(tile0 * 32);
  __syncthreads();                                   // the twiddle table
  for (int it = 0; it < n_tiles; ++it) {
    const int col0 = (tile0 + it) * 32;
    cd v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = (t + 16 * j) * ld + col0 + 2 * s;
      v[j] = make_double2(n >= n_lead && n < n_valid ? (double)ra[j] - mean : 0.0,
                          n + 1 >= n_lead && n + 1 < n_valid ? (double)rb[j] - mean : 0.0);
      if (__any(seg_in_ramp(ramp, n) || seg_in_ramp(ramp, n + 1)))      // wave-uniform: the ramps are rare
        v[j] = make_double2(v[j].x * seg_weight(ramp, n), v[j].y * seg_weight(ramp, n + 1));
      acc += v[j].x + v[j].y;
    }
    if (it + 1 < n_tiles) fetch(col0 + 32);          // in flight during this tile's transform
    d_fft256(v, twl + t, ex_re + s * kDCol, ex_im + s * kDCol, t);
    __syncthreads();                                 // the tile aliases the planes
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[(t + 16 * j) * 17 + s] = v[j];
    __syncthreads();
    // split and twiddle: thread = one real column, rows k = k0 + 8 i
    const int cr = tid & 31, k0 = tid >> 5, m = cr >> 1;
    const bool odd = cr & 1;
    const int64_t col = col0 + cr;
    cd w = d_phase(tw_hi, tw_lo, col * k0, lg_p);
    const cd st = d_phase(tw_hi, tw_lo, col * 8, lg_p);
    cd* o = out + (int64_t)c * out_cstride + col;
    for (int k = k0; k <= 128; k += 8) {
      const cd zk = tile[k * 17 + m], zm = tile[((256 - k) & 255) * 17 + m];
      const cd val = odd ? make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x))
                         : make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
      o[(int64_t)k * ld] = dmul(val, w);
      if (rows_out > 129 && k > 0 && k < 128)          // Y0[256 - k] = conj(Y0[k]) for a real column
        o[(int64_t)(256 - k) * ld] = dmul(dconj(val), d_phase(tw_hi, tw_lo, col * (256 - k), lg_p));
      w = dmul(w, st);
    }
    __syncthreads();                                 // the tile is read before the next exchange overwrites it
  }
  if (fold_parts) fold_store(acc, ex_re, fold_parts, segs.n_channels, ch, tid);      // (kernel-uniform)
}

// ---------------------------------------------------------------------------
// Pass A, len = 256 q, q = 2 or 4 (P = 2^21, 2^22): two real subsequences of a column per FFT256
// (kernels.hip: k_fft_colsq_real2).  grid (ld / 16, slots); dynamic LDS kFwd64Lds (+ one tile for q = 4)
// ---------------------------------------------------------------------------
template <int LQ>
__global__ void __launch_bounds__(256) k_fwd64_colsq_real2(const float* __restrict__ in, cd* __restrict__ out,
                                                           int ld, int64_t in_cstride, int64_t out_cstride,
                                                           int lg_p, const cd* __restrict__ tw_hi,
                                                           const cd* __restrict__ tw_lo,
                                                           const double* __restrict__ sums, double inv_n,
                                                           const SegIn segs, int rows_out, int in_stride, int in_offset,
                                                           double* __restrict__ fold_parts) {
  // in_stride A, in_offset a: the transform of the subsequence x[A n + a] (long mode: planner.h, EpochPlan::long_a)
  constexpr int q = 1 << LQ, len = 256 * q, np = q / 2;
  static_assert(q == 2 || q == 4, "two or four subsequences");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* const ex_re = reinterpret_cast<double*>(smem);
  double* const ex_im = ex_re + 16 * kDCol;
  cd* const planes = reinterpret_cast<cd*>(smem);                        // last pair's tile aliases the planes
  cd* const twl = reinterpret_cast<cd*>(smem + 2 * kDPlaneBytes);
  cd* const tile0 = twl + 256;                                           // first pair's tile (q = 4)
  const int c = blockIdx.y, col0 = blockIdx.x * 16, tid = threadIdx.x;
  const int s = tid & 15, t = tid >> 4;
  d_fill_twl(twl, tw_hi, tid);
  const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
  const int n_valid = (int)segs.n_valid[g], n_lead = (int)segs.n_lead[g];      // below P <= 2^24
  const SegRamp ramp = seg_ramp(segs, g);
  const float* x = in + (int64_t)ch * in_cstride + segs.x_off[g];
  const double mean = fold_parts ? 0.0 : sums[ch] * inv_n;     // (fold_parts: k_fwd64_cols256_real2)
  double acc = 0.0;
  const int top = n_valid > 0 ? n_valid - 1 : 0;
  cd v[16], u[np > 1 ? 16 : 1];
#pragma unroll
  for (int p = 0; p < np; ++p) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int na = ((q * (t + 16 * j) + 2 * p) * ld + col0 + s) * in_stride + in_offset, nb = na + ld * in_stride;
      const float xa = x[min(max(na, n_lead), top)], xb = x[min(max(nb, n_lead), top)];   // clamped
      v[j] = make_double2(na >= n_lead && na < n_valid ? (double)xa - mean : 0.0,
                          nb >= n_lead && nb < n_valid ? (double)xb - mean : 0.0);
    }
    // the faded edges of a time block, after every load has been used (a branch between the loads would stop the
    // compiler from issuing them together: +40 % on this one-wave-per-SIMD kernel): wave-uniform per row, rare
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int na = ((q * (t + 16 * j) + 2 * p) * ld + col0 + s) * in_stride + in_offset, nb = na + ld * in_stride;
      if (__any(seg_in_ramp(ramp, na) || seg_in_ramp(ramp, nb)))
        v[j] = make_double2(v[j].x * seg_weight(ramp, na), v[j].y * seg_weight(ramp, nb));
      acc += v[j].x + v[j].y;
    }
    __syncthreads();                      // twiddle table written / previous exchange read
    d_fft256(v, twl + t, ex_re + s * kDCol, ex_im + s * kDCol, t);
    if (p < np - 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { tile0[(t + 16 * j) * 17 + s] = v[j]; u[j] = v[j]; }
    }
  }
  __syncthreads();                        // the last pair's tile aliases the planes
#pragma unroll
  for (int j = 0; j < 16; ++j) planes[(t + 16 * j) * 17 + s] = v[j];
  __syncthreads();
  cd* o = out + (int64_t)c * out_cstride + col0 + s;
  const int64_t col = col0 + s;
  const cd stq = d_phase(tw_hi, tw_lo, col * 256, lg_p), st16 = d_phase(tw_hi, tw_lo, col * 16, lg_p);
  cd wj = d_phase(tw_hi, tw_lo, col * t, lg_p);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kb = t + 16 * j, km = (256 - kb) & 255;
    cd w[q];
#pragma unroll
    for (int p = 0; p < np; ++p) {
      const cd zk = p < np - 1 ? u[j] : v[j];
      const cd zm = p < np - 1 ? tile0[km * 17 + s] : planes[km * 17 + s];
      const cd xa = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
      const cd xb = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
      // W_len^(-a kb), a = 2p, 2p + 1: index a kb (4096 / len) of the 4096 table
      w[2 * p] = p == 0 ? xa : dmul(xa, tw_hi[(2 * p * kb * (kRowLenDev / len)) & 4095]);
      w[2 * p + 1] = dmul(xb, tw_hi[((2 * p + 1) * kb * (kRowLenDev / len)) & 4095]);
    }
    d_dft_small<q>(w);
    cd ph = wj;
#pragma unroll
    for (int ka = 0; ka < q; ++ka) {
      const int k = kb + 256 * ka;
      if (k < rows_out) o[(int64_t)k * ld] = dmul(w[ka], ph);
      ph = dmul(ph, stq);
    }
    wj = dmul(wj, st16);
  }
  if (fold_parts) {                       // (kernel-uniform)
    __syncthreads();                      // the tiles in the planes are read
    fold_store(acc, ex_re, fold_parts, segs.n_channels, ch, tid);
  }
}

// ---------------------------------------------------------------------------
// Pass A, any len = 1 .. 128 (P = 2^12 .. 2^19): radix-2 in LDS, 16 columns per workgroup
// (kernels.hip: k_fft_cols<-1, true>).  grid (ld / 16, slots), dynamic LDS len * 16 * 16 bytes
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fwd64_cols_small(const float* __restrict__ in, cd* __restrict__ out,
                                                          int len, int log2len, int ld, int64_t in_cstride,
                                                          int64_t out_cstride, int lg_p,
                                                          const cd* __restrict__ tw_hi, const cd* __restrict__ tw_lo,
                                                          const double* __restrict__ sums, double inv_n,
                                                          const SegIn segs, int rows_out, double* __restrict__ fold_parts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* buf = reinterpret_cast<cd*>(smem);
  const int c = blockIdx.y, col0 = blockIdx.x * 16, total = len * 16;
  const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
  const int64_t n_valid = segs.n_valid[g], n_lead = segs.n_lead[g];
  const SegRamp ramp = seg_ramp(segs, g);
  const float* x = in + (int64_t)ch * in_cstride + segs.x_off[g];
  const double mean = fold_parts ? 0.0 : sums[ch] * inv_n;     // (fold_parts: k_fwd64_cols256_real2)
  double acc = 0.0;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int i = e >> 4, cc = e & 15;
    const int64_t n = (int64_t)i * ld + col0 + cc;
    const double val = n >= n_lead && n < n_valid ? ((double)x[n] - mean) * seg_weight(ramp, (int)n) : 0.0;
    acc += val;
    buf[e] = make_double2(val, 0.0);
  }
  __shared__ double fold_scratch[4];
  if (fold_parts) fold_store(acc, fold_scratch, fold_parts, segs.n_channels, ch, threadIdx.x);     // (kernel-uniform)
  __syncthreads();
  const int half_total = 16 * (len >> 1);
  for (int h = len >> 1, st = 0; h >= 1; h >>= 1, ++st) {          // decimation in frequency
    for (int b = threadIdx.x; b < half_total; b += 256) {
      const int f = b & 15, j = b >> 4;
      const int pos = j & (h - 1);
      const int i0 = ((j - pos) << 1) + pos;
      cd* p0 = buf + i0 * 16 + f;
      cd* p1 = p0 + h * 16;
      const cd a = *p0, cc = *p1;
      const cd w = tw_hi[(pos << st) * (kRowLenDev >> log2len)];
      *p0 = dadd(a, cc);
      *p1 = dmul(dsub(a, cc), w);
    }
    __syncthreads();
  }
  cd* o = out + (int64_t)c * out_cstride;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int k = e >> 4, cc = e & 15;
    if (k >= rows_out) continue;                     // (real input: the caller may keep rows 0 .. len/2 only)
    const int kr = log2len ? (int)(__brev((unsigned)k) >> (32 - log2len)) : 0;
    cd v = buf[kr * 16 + cc];
    if (len > 1) v = dmul(v, d_phase(tw_hi, tw_lo, (int64_t)k * (col0 + cc), lg_p));
    o[(int64_t)k * ld + col0 + cc] = v;
  }
}

// ---------------------------------------------------------------------------
// Pass B: one 4096-point row per workgroup -- FFT256 over the 16 stride-16 subsequences, twiddle
// W_4096^(-kb a), DFT16 over a (kernels.hip: rows_fast_body<-1, 4>) -- float64 in, float32 out.
//   mirror > 0: outputs below 2048 go to row `row`, the upper half as the reflected lower half of
//   row mirror - row (spectrum of a real signal, k1-major); mirror = 0: outputs below out_len.
// grid (n_rows, slots), dynamic LDS kFwd64Lds
// ---------------------------------------------------------------------------
__device__ __forceinline__ int dpad(int i) { return i + (i >> 4); }

// precision = auto (detect.hip): the band energies of the spectrum are summed where it is made.  A wave of the row
// pass holds 64 consecutive k2 of one row k1 (bins k1 + p1 k2), ascending or -- the reflected half -- descending, and
// from k2 = 16 on the band of a bin is the band of its k2 (sixteen per octave: kernels.h, spec_band; p1 is a power of
// two), 2^(e - 4) consecutive k2 of octave e: aligned groups of 1 .. 64 lanes, summed by a butterfly in a fixed order
// and stored once by the group's first k2 -- no atomics, no staging, the same bits every run.  k2 < 16 (where the
// band depends on k1 too) are kept bin by bin.  Row layout (kernels.h: kRowBands): [0, 16) those bins, [16, 128) the
// bands of k2 = 16 .. 2047.  The butterfly is vector ALU only (DPP within rows of 16 lanes, v_permlane16_swap /
// v_permlane32_swap across them): through ds_bpermute (__shfl_xor) the 82 exchanges per thread doubled the kernel's
// LDS instructions and cost more than the separate pass over the spectrum they replace (0.15 against 0.12 ms).
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// own value + the value of the lane 16 (32) away: v_permlane16_swap (v_permlane32_swap) exchanges the odd rows (upper
// half) of one register with the even rows (lower half) of another, so two copies of e come back as {own, partner} in
// some order.  Through the builtin hipcc 7.2 takes the two results of a swap of equal operands for one value (it emits
// e' + e'), hence the instruction itself.
__device__ __forceinline__ float swap_sum16(float e) {
  float a = e, b = e;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // (two wait states after the VALU that wrote them)
  return a + b;
}
__device__ __forceinline__ float swap_sum32(float e) {
  float a = e, b = e;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__host__ __device__ constexpr int band_group_log2(int k2) {      // log2 of the run of k2 that shares k2's band
  int e = 0;
  while ((2 << e) <= k2) ++e;                                     // floor(log2 k2), k2 >= 1
  return k2 < 32 ? 0 : e - 4;
}
// D_MAX: the wave's largest group (log2), known at compile time except for k2 < 256 (D_MAX = -1: from the wave's top k2)
template <int D_MAX>
__device__ __forceinline__ void row_band_sums(float e, int k2, float* __restrict__ hrow) {
  const int depth = k2 < 32 ? 0 : 27 - __clz(k2);
  int d_max = D_MAX;
  if (D_MAX < 0) {
    const int top = __builtin_amdgcn_readfirstlane(k2 | 63);
    d_max = top < 32 ? 0 : min(3, 27 - __clz(top));     // (k2 < 256: groups of eight at most)
  }
  if (d_max >= 1) {                                    // (the one step a wave's lanes may differ in: k2 < 32 keep their own)
    const float o = dpp_get<0xB1>(e);                  // quad_perm [1, 0, 3, 2]: lane ^ 1
    e = depth >= 1 ? e + o : e;
  }
  if (d_max >= 2) e += dpp_get<0x4E>(e);               // quad_perm [2, 3, 0, 1]: lane ^ 2
  if (d_max >= 3) e += dpp_get<0x141>(e);              // row_half_mirror: the other quad of the eight (quads are uniform by now)
  if (d_max >= 4) e += dpp_get<0x140>(e);              // row_mirror: the other eight of the row
  if (d_max >= 5) {                                    // the neighbouring row of 16
    e = swap_sum16(e);
  }
  if (d_max >= 6) {                                    // the other half of the wave
    e = swap_sum32(e);
  }
  if ((k2 & ((1 << depth) - 1)) == 0)
    hrow[k2 < 16 ? k2 : 16 + (int)(__float_as_uint((float)k2) >> 19) - (127 * 16 + 64)] = e;
}

__device__ __forceinline__ void constexpr_band_sums(int ka, float e, int idx, int row, int mirror, float* __restrict__ hslot) {
  // ka -> the wave's largest group at compile time: 256 .. 511 -> 16 lanes, 512 .. 1023 -> 32, from 1024 on the whole wave
  const bool up = ka >= 8;
  if (up && !(mirror > 0 && row > 0 && 2 * row < mirror)) return;                  // workgroup-uniform
  const int kq = up ? 15 - ka : ka;                                                 // which 256 of the destination row
  const int k2 = up ? kRowLenDev - 1 - idx : idx;
  float* const hrow = hslot + (int64_t)(up ? mirror - row : row) * kRowBands;
  if (kq == 0) row_band_sums<-1>(e, k2, hrow);
  else if (kq == 1) row_band_sums<4>(e, k2, hrow);
  else if (kq <= 3) row_band_sums<5>(e, k2, hrow);
  else row_band_sums<6>(e, k2, hrow);
}

// comb_n > 1 (long mode): this is subsequence comb_a of comb_n; every output bin k is multiplied by W_pt^(a k),
// pt = 2^lg_pt the true FFT length, and added to what the earlier subsequences left (a = 0 stores).
// HIST / FOLD: instantiations with and without the band sums and the mean's removal (as run-time switches inside one
// kernel they cost the plain row pass 25 registers and 0.06 ms: the sixteen stores of the epilogue ended up in sixteen
// basic blocks)
template <bool HIST, bool FOLD>
__global__ void __launch_bounds__(256) k_fwd64_rows(const cd* __restrict__ in, float2* __restrict__ out,
                                                    int64_t in_cstride, int64_t out_cstride,
                                                    const cd* __restrict__ tw_hi, int out_len, int mirror,
                                                    int comb_a, int comb_n, int lg_pt, float* __restrict__ hist,
                                                    int hist_rows, const double* __restrict__ fold_sums, double inv_n,
                                                    int n_valid, int lg_p, int lg_p1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* const ex_re = reinterpret_cast<double*>(smem);
  double* const ex_im = ex_re + 16 * kDCol;
  cd* const buf = reinterpret_cast<cd*>(smem);                           // 4096 (+256 pad) elements, aliases the planes
  cd* const twl = reinterpret_cast<cd*>(smem + 2 * kDPlaneBytes);
  const int tid = threadIdx.x, a = tid & 15, t = tid >> 4, row = blockIdx.x;
  const cd* x = in + (int64_t)blockIdx.y * in_cstride + (int64_t)row * kRowLenDev;
  float2* o = out + (int64_t)blockIdx.y * out_cstride;
  d_fill_twl(twl, tw_hi, tid);
  cd v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = x[16 * (t + 16 * j) + a];
  if (FOLD) {
    // The column pass transformed x, not x - mean (kernels.h: launch_fwd64_cols, fold_mean).  The mean's share of
    // Y[k1][n2] is mean W_P^(-n2 k1) G_c(k1), G_c(k1) = sum_{n1 < c} W_P1^(-n1 k1) a geometric sum over the c samples
    // column n2 holds (c = ceil(N / 4096) for the first columns, one less for the rest): taken out here, in float64,
    // before the row's transform -- transforms.py:142-143's x - mean(x) without a pass of its own over x.
    const cd* const tw_lo = tw_hi + 4096;
    const double mean = fold_sums[blockIdx.y] * inv_n;
    const int c_hi = (n_valid + kRowLenDev - 1) / kRowLenDev, n_split = n_valid - (c_hi - 1) * kRowLenDev;
    cd g_hi = make_double2((double)c_hi, 0.0), g_lo = make_double2((double)(c_hi - 1), 0.0);
    if (row > 0) {
      const cd w1 = d_phase(tw_hi, tw_lo, row, lg_p1);
      const cd den = make_double2(1.0 - w1.x, -w1.y);
      const double r2 = 1.0 / (den.x * den.x + den.y * den.y);
      const cd inv = make_double2(den.x * r2, -den.y * r2);
      const cd wh = d_phase(tw_hi, tw_lo, (int64_t)row * c_hi, lg_p1), wl = d_phase(tw_hi, tw_lo, (int64_t)row * (c_hi - 1), lg_p1);
      g_hi = dmul(make_double2(1.0 - wh.x, -wh.y), inv);
      g_lo = dmul(make_double2(1.0 - wl.x, -wl.y), inv);
    }
    g_hi = make_double2(g_hi.x * mean, g_hi.y * mean);
    g_lo = make_double2(g_lo.x * mean, g_lo.y * mean);
    cd w = d_phase(tw_hi, tw_lo, (int64_t)row * (a + 16 * t), lg_p);
    const cd step = d_phase(tw_hi, tw_lo, (int64_t)row * 256, lg_p);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n2 = a + 16 * t + 256 * j;
      v[j] = dsub(v[j], dmul(n2 < n_split ? g_hi : g_lo, w));
      w = dmul(w, step);
    }
  }
  __syncthreads();
  d_fft256(v, twl + t, ex_re + a * kDCol, ex_im + a * kDCol, t);
  __syncthreads();                       // the element buffer aliases the exchange planes
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kb = t + 16 * j;
    buf[dpad(16 * kb + a)] = dmul(v[j], tw_hi[(kb * a) & 4095]);
  }
  __syncthreads();
  const int kb = tid;
  cd u[16];
#pragma unroll
  for (int aa = 0; aa < 16; ++aa) u[aa] = buf[dpad(16 * kb + aa)];
  d_dft16(u);
#pragma unroll
  for (int ka = 0; ka < 16; ++ka) {
    const int idx = kb + 256 * ka;
    if (comb_n > 1) {                    // (long mode implies the reflected layout: mirror = P1)
      const cd* tw_lo = tw_hi + 4096;
      int drow, didx;
      cd v2 = u[ka];
      if (idx < kRowLenDev / 2) { drow = row; didx = idx; }
      else if (row > 0 && 2 * row < mirror) { drow = mirror - row; didx = kRowLenDev - 1 - idx; v2 = dconj(v2); }
      else continue;
      const int64_t kdest = (int64_t)drow + (int64_t)mirror * didx;
      v2 = dmul(v2, d_phase(tw_hi, tw_lo, (int64_t)comb_a * kdest, lg_pt));
      float2* dst = o + (int64_t)drow * kRowLenDev + didx;
      if (comb_a == 0) *dst = make_float2((float)v2.x, (float)v2.y);
      else { const float2 old = *dst; *dst = make_float2((float)((double)old.x + v2.x), (float)((double)old.y + v2.y)); }
      continue;
    }
    const float2 val = make_float2((float)u[ka].x, (float)u[ka].y);
    if (HIST) {                          // |X|^2 of the rounded bin into its row's band sums
      float* const hslot = hist + (int64_t)blockIdx.y * hist_rows * kRowBands;
      const float e = val.x * val.x + val.y * val.y;
      // (ka is a constant of the unrolled loop: k2 = kb + 256 ka, reflected k2 = 255 - kb + 256 (15 - ka))
      constexpr_band_sums(ka, e, idx, row, mirror, hslot);
    }
    if (mirror == 0) {
      if (256 * ka < out_len) o[(int64_t)row * kRowLenDev + idx] = val;
    } else if (idx < kRowLenDev / 2) {
      o[(int64_t)row * kRowLenDev + idx] = val;
    } else if (row > 0 && 2 * row < mirror) {
      o[(int64_t)(mirror - row) * kRowLenDev + (kRowLenDev - 1 - idx)] = make_float2(val.x, -val.y);
    }
  }
}

// ---------------------------------------------------------------------------
// Block convolution, forward half (kernels.h: BcBlocks): the 4096-point transforms of two blocks of one channel,
// pass B's arithmetic on the recording itself -- x - mean in float64, zero outside the epoch -- rounded to
// float32 per bin like the spectrum of the segment transforms.  The blocks are real: blocks 2 i and 2 i + 1 ride
// one transform as z = x' + i x'', and X'[k] = (Z[k] + conj Z[-k]) / 2, X''[k] = (Z[k] - conj Z[-k]) / 2i come
// apart through LDS afterwards.  grid (ceil(blocks / 2), channels), dynamic LDS kFwd64Lds
// ---------------------------------------------------------------------------
struct BcWindow { int64_t in0, e0, e1; };
__device__ __forceinline__ BcWindow bc_window(const BcBlocks& bl, int blk) {
  int e = 0;
  while (e + 1 < bl.n_epochs && blk >= bl.blk_first[e + 1]) ++e;
  BcWindow w;
  w.in0 = (((bl.g_lo[e] / bl.hop) & ~(int64_t)1) + (blk - bl.blk_first[e])) * bl.hop - bl.back;   // kernels.h: BcBlocks
  w.e0 = bl.epoch_start[e];
  w.e1 = bl.epoch_stop[e];
  return w;
}

__global__ void __launch_bounds__(256) k_bc_forward(const float* __restrict__ x, float2* __restrict__ xb,
                                                    const BcBlocks bl, int blk0, int nblk, int64_t n_samples,
                                                    const cd* __restrict__ tw_hi, const double* __restrict__ sums,
                                                    double inv_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* const ex_re = reinterpret_cast<double*>(smem);
  double* const ex_im = ex_re + 16 * kDCol;
  cd* const buf = reinterpret_cast<cd*>(smem);
  cd* const twl = reinterpret_cast<cd*>(smem + 2 * kDPlaneBytes);
  const int tid = threadIdx.x, a = tid & 15, t = tid >> 4, ch = blockIdx.y;
  const int lb = 2 * blockIdx.x;
  const bool second = lb + 1 < nblk;
  const BcWindow w0 = bc_window(bl, blk0 + lb), w1 = bc_window(bl, blk0 + (second ? lb + 1 : lb));
  const float* xc = x + (int64_t)ch * n_samples;
  const double mean = sums[ch] * inv_n;
  d_fill_twl(twl, tw_hi, tid);
  cd v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int off = 16 * (t + 16 * j) + a;
    const int64_t n = w0.in0 + off, m = w1.in0 + off;
    const float xa = xc[min(max(n, w0.e0), w0.e1 - 1)];       // clamped: no branch around the loads
    const float xm = xc[min(max(m, w1.e0), w1.e1 - 1)];
    v[j] = make_double2(n >= w0.e0 && n < w0.e1 ? (double)xa - mean : 0.0,
                        second && m >= w1.e0 && m < w1.e1 ? (double)xm - mean : 0.0);
  }
  if (bl.ramp > 0) {
    // precision = exact: the block's edges fade (C2), so that a strong line elsewhere in the band does not reach this
    // scale's bins as the leakage of a cut -- which float32 could not cancel again after the product
    const double inv = 1.0 / (double)bl.ramp;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int off = 16 * (t + 16 * j) + a;
      const int edge = min(off, kRowLenDev - 1 - off);
      if (edge < bl.ramp) {
        const double u = ((double)edge + 0.5) * inv, w = u * u * u * (10.0 + u * (6.0 * u - 15.0));
        v[j] = make_double2(v[j].x * w, v[j].y * w);
      }
    }
  }
  __syncthreads();
  d_fft256(v, twl + t, ex_re + a * kDCol, ex_im + a * kDCol, t);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kb = t + 16 * j;
    buf[dpad(16 * kb + a)] = dmul(v[j], tw_hi[(kb * a) & 4095]);
  }
  __syncthreads();
  cd u[16];
#pragma unroll
  for (int aa = 0; aa < 16; ++aa) u[aa] = buf[dpad(16 * tid + aa)];
  d_dft16(u);
  __syncthreads();                        // everyone has read the buffer: Z goes into it in natural order
#pragma unroll
  for (int ka = 0; ka < 16; ++ka) buf[tid + 256 * ka] = u[ka];
  __syncthreads();
  float2* o0 = xb + ((int64_t)lb * bl.n_channels + ch) * kRowLenDev;
  float2* o1 = o0 + (int64_t)bl.n_channels * kRowLenDev;
#pragma unroll
  for (int ka = 0; ka < 16; ++ka) {
    const int k = tid + 256 * ka;
    const cd z = u[ka], zp = buf[(kRowLenDev - k) & (kRowLenDev - 1)];
    o0[k] = make_float2((float)(0.5 * (z.x + zp.x)), (float)(0.5 * (z.y - zp.y)));
    if (second) o1[k] = make_float2((float)(0.5 * (z.y + zp.y)), (float)(0.5 * (zp.x - z.x)));
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int ilog2_64(int64_t v) { int l = 0; while (((int64_t)1 << l) < v) ++l; return l; }

template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes) {
  return bytes > 48 * 1024 ? hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
                           : hipSuccess;
}

// tables: [0, 4096) W_4096^a, [4096, 8192) W_2^24^b
void fwd64_fill_tables(double2* host) {
  for (int a = 0; a < 4096; ++a) {
    const double ang = -2.0 * M_PI * (double)a / 4096.0;
    host[a] = make_double2(std::cos(ang), std::sin(ang));
  }
  // exact at the eighths of the circle (cos / sin of float64 multiples of pi are not)
  host[0] = make_double2(1.0, 0.0); host[1024] = make_double2(0.0, -1.0);
  host[2048] = make_double2(-1.0, 0.0); host[3072] = make_double2(0.0, 1.0);
  for (int b = 0; b < 4096; ++b) {
    const double ang = -2.0 * M_PI * (double)b / 16777216.0;
    host[4096 + b] = make_double2(std::cos(ang), std::sin(ang));
  }
}

hipError_t launch_fwd64_cols(const float* in, double2* y, int p1, int64_t in_cstride, int64_t y_cstride,
                             int64_t p, const double2* tables, const double* sums, double inv_n,
                             const SegIn& segs, int n_segments, int rows_out, hipStream_t st, int in_stride,
                             int in_offset, bool fold_mean) {
  const int slots = segs.n_channels * n_segments, ld = kRowLenDev, lg_p = ilog2_64(p);
  const cd* tw_hi = tables;
  const cd* tw_lo = tables + 4096;
  if (lg_p > 24 || ((int64_t)p1 * kRowLenDev) != p || rows_out < 1 || rows_out > p1 ||
      (int64_t)rows_out * kRowLenDev > y_cstride) return hipErrorInvalidValue;     // y holds rows_out rows per slot
  if (in_stride < 1 || in_offset < 0 || in_offset >= in_stride || (in_stride > 1 && p1 != 512 && p1 != 1024))
    return hipErrorInvalidValue;                     // strided input: the 2^21 / 2^22-point kernels only (long mode)
  hipError_t e;
  static_assert((kRowLenDev / 32 + kColsTiles - 1) / kColsTiles == fold_parts(256) && kRowLenDev / 16 == fold_parts(512) &&
                fold_parts(512) <= kSumParts, "partial sums of the folded mean: one per workgroup of the column pass");
  if (fold_mean && (n_segments != 1 || in_stride != 1 || segs.n_lead[0] != 0 || segs.ramp_lo[0] != 0 ||
                    segs.ramp_hi[0] != 0 || segs.x_off[0] != 0)) return hipErrorInvalidValue;
  double* const parts = fold_mean ? const_cast<double*>(sums) : nullptr;
  if (p1 == 256) {
    if ((e = allow_lds(k_fwd64_cols256_real2, kFwd64Lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fwd64_cols256_real2, dim3((ld / 32 + kColsTiles - 1) / kColsTiles, slots), dim3(256), kFwd64Lds, st, in, y, ld,
                       in_cstride, y_cstride, lg_p, tw_hi, tw_lo, sums, inv_n, segs, rows_out, parts);
  } else if (p1 == 512) {
    if ((e = allow_lds(k_fwd64_colsq_real2<1>, kFwd64Lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fwd64_colsq_real2<1>, dim3(ld / 16, slots), dim3(256), kFwd64Lds, st, in, y, ld,
                       in_cstride, y_cstride, lg_p, tw_hi, tw_lo, sums, inv_n, segs, rows_out, in_stride, in_offset, parts);
  } else if (p1 == 1024) {
    const size_t lds = kFwd64Lds + 256 * 17 * sizeof(cd);
    if ((e = allow_lds(k_fwd64_colsq_real2<2>, lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fwd64_colsq_real2<2>, dim3(ld / 16, slots), dim3(256), lds, st, in, y, ld,
                       in_cstride, y_cstride, lg_p, tw_hi, tw_lo, sums, inv_n, segs, rows_out, in_stride, in_offset, parts);
  } else if (p1 >= 1 && p1 <= 128) {
    const size_t lds = (size_t)p1 * 16 * sizeof(cd);
    hipLaunchKernelGGL(k_fwd64_cols_small, dim3(ld / 16, slots), dim3(256), lds, st, in, y, p1, ilog2_64(p1), ld,
                       in_cstride, y_cstride, lg_p, tw_hi, tw_lo, sums, inv_n, segs, rows_out, parts);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_fwd64_rows(const double2* y, float2* x, int n_rows, int64_t y_cstride, int64_t x_cstride,
                             const double2* tables, int n_slots, int out_len, int mirror, hipStream_t st,
                             int comb_a, int comb_n, int64_t p_true, float* hist, int hist_rows,
                             const double* fold_sums, double inv_n, int64_t n_valid, int p1) {
  if (fold_sums && (comb_n > 1 || p1 < 1 || (p1 & (p1 - 1)) || n_valid < 1 || n_valid > (int64_t)p1 * kRowLenDev)) return hipErrorInvalidValue;
  if (comb_n > 1 && (mirror == 0 || comb_a < 0 || comb_a >= comb_n || ilog2_64(p_true) > 24)) return hipErrorInvalidValue;
  // band sums: every row k1 of the spectrum's positive half once -- from its own workgroup or, reflected, from its twin's
  if (hist && (comb_n > 1 || out_len < kRowLenDev / 2 || hist_rows != (mirror > 0 ? mirror : n_rows))) return hipErrorInvalidValue;
  hipError_t e = hipSuccess;
#define GCWT_ROWS(H, F)                                                                                                  \
  do {                                                                                                                   \
    if ((e = allow_lds(k_fwd64_rows<H, F>, kFwd64Lds)) != hipSuccess) return e;                                          \
    hipLaunchKernelGGL((k_fwd64_rows<H, F>), dim3(n_rows, n_slots), dim3(256), kFwd64Lds, st, y, x, y_cstride, x_cstride, \
                       tables, out_len, mirror, comb_a, comb_n, comb_n > 1 ? ilog2_64(p_true) : 0, hist, hist_rows,      \
                       fold_sums, inv_n, (int)n_valid, fold_sums ? ilog2_64((int64_t)p1 * kRowLenDev) : 0,                \
                       fold_sums ? ilog2_64(p1) : 0);                                                                    \
  } while (0)
  if (hist && fold_sums) GCWT_ROWS(true, true);
  else if (hist) GCWT_ROWS(true, false);
  else if (fold_sums) GCWT_ROWS(false, true);
  else GCWT_ROWS(false, false);
#undef GCWT_ROWS
  return hipGetLastError();
}

hipError_t launch_bc_forward(const float* x, float2* xb, const BcBlocks& bl, int blk0, int nblk, int64_t n_samples,
                             const double2* tables, const double* sums, double inv_n, hipStream_t st) {
  if (nblk <= 0) return hipSuccess;
  if (bl.n_epochs < 1 || bl.n_epochs > kSegBatch || bl.hop < 1 || bl.n_channels < 1 || bl.n_channels > 65535 ||
      blk0 < 0 || blk0 + nblk > bl.blk_first[bl.n_epochs])
    return hipErrorInvalidValue;
  hipError_t e = allow_lds(k_bc_forward, kFwd64Lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bc_forward, dim3((nblk + 1) / 2, bl.n_channels), dim3(256), kFwd64Lds, st, x, xb, bl, blk0,
                     nblk, n_samples, tables, sums, inv_n);
  return hipGetLastError();
}

}  // namespace gcwt
