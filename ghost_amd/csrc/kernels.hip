// kernels.hip -- gfx950 device code of the CWT engine and its launch wrappers.
//
// Stages:
//   k_channel_sum      per-channel sum for the global mean     (transforms.py:143)
//   k_build_bank       Morse one-sided filter bank, fp64 -> fp32 (morseutils.py:115-131)
//   k_build_direct     literal L-tap kernels for scales that reach Nyquist
//                                                              (morseutils.py:117-149)
//   k_fft_cols/rows    two-pass radix-2 LDS FFT used for the per-epoch forward
//                      FFT and the per-level inverse FFTs (replaces the per-chunk
//                      FFTs of convolution.py:74-76)
//   k_block_fft        256-point spectra of overlapping decimated blocks
//   k_synth            filter * polyphase twiddle * 256-point IFFT * |.| * store
//                                                              (transforms.py:203-204)
//   k_direct           time-domain convolution for direct scales
//   k_zero_gaps        zero samples outside every epoch        (transforms.py:185)
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "morse_exact.h"
#include "synth_math.h"

namespace gcwt {

typedef float2 cf;

__device__ __forceinline__ cf cmul(cf a, cf b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by SIGN * i
template <int SIGN>
__device__ __forceinline__ cf mul_si(cf a) {
  return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// low cut of a level's spectrum slice (kernels.h: RowTaper): factor for element i of row `row`
__device__ __forceinline__ float row_taper(const RowTaper& tp, int row, int i) {
  const float u = ((float)(row + tp.p1 * i) - tp.k0) * tp.inv_width;
  return u >= 1.f ? 1.f : (u <= 0.f ? 0.f : 0.5f - 0.5f * cospif(u));
}

// ---------------------------------------------------------------------------
// 16-point DFT in registers, natural order in and out.
//   X[k] = sum_n x[n] exp(SIGN 2 pi i n k / 16)
// n = n1 + 4 n2, k = 4 k1 + k2: DFT4 over n2, twiddle W16^(n1 k2), DFT4 over n1.
// ---------------------------------------------------------------------------
template <int SIGN>
__device__ __forceinline__ void dft4(cf& a, cf& b, cf& c, cf& d) {
  cf s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = mul_si<SIGN>(csub(b, d));
  a = cadd(s0, s2);
  c = csub(s0, s2);
  b = cadd(s1, s3);
  d = csub(s1, s3);
}

template <int SIGN>
__device__ __forceinline__ void dft16(cf v[16]) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;  // cos, sin(pi/8)
  const float h = 0.70710678118654752f;
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) dft4<SIGN>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
  // v[n1 + 4 k2] *= W16^(n1 k2)
  const float sg = (float)SIGN;
  v[5] = cmul(v[5], make_float2(c1, sg * s1));     // q = 1
  v[9] = cmul(v[9], make_float2(h, sg * h));       // q = 2
  v[13] = cmul(v[13], make_float2(s1, sg * c1));   // q = 3
  v[6] = cmul(v[6], make_float2(h, sg * h));       // q = 2
  v[10] = mul_si<SIGN>(v[10]);                     // q = 4
  v[14] = cmul(v[14], make_float2(-h, sg * h));    // q = 6
  v[7] = cmul(v[7], make_float2(s1, sg * c1));     // q = 3
  v[11] = cmul(v[11], make_float2(-h, sg * h));    // q = 6
  v[15] = cmul(v[15], make_float2(-c1, -sg * s1)); // q = 9
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2)
    dft4<SIGN>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
  // now v[k1 + 4 k2] holds X[4 k1 + k2]: transpose the 4x4 index
  cf t;
#define GCWT_SWAP(a, b) t = v[a]; v[a] = v[b]; v[b] = t;
  GCWT_SWAP(1, 4) GCWT_SWAP(2, 8) GCWT_SWAP(3, 12) GCWT_SWAP(6, 9) GCWT_SWAP(7, 13) GCWT_SWAP(11, 14)
#undef GCWT_SWAP
}

// ---------------------------------------------------------------------------
// 256-point FFT by 16 threads (one "column"), 16 points per thread.
//   in : v[j] = in[t + 16 j]       out: v[j] = out[t + 16 j]
// k = k1 + 16 k2 (k1 = t), m = 16 m1 + m2:  DFT16 over k2, twiddle W256^(k1 m2),
// exchange through LDS, DFT16 over k1.  tw[m2] = exp(SIGN 2 pi i t m2 / 256).
// ex_re/ex_im: this column's 16 x 17 float planes.  Contains one __syncthreads.
// ---------------------------------------------------------------------------
constexpr int kExPitch = 17;
constexpr int kExCol = 16 * kExPitch;  // floats per column per plane
constexpr int kExColD = 290;           // same, when the lanes of a wave run over columns (see k_fft_cols256)

template <int SIGN>
__device__ __forceinline__ void fft256_16t(cf v[16], const cf tw[16], float* ex_re, float* ex_im,
                                           int t) {
  dft16<SIGN>(v);
#pragma unroll
  for (int m2 = 0; m2 < 16; ++m2) {
    cf u = cmul(v[m2], tw[m2]);
    ex_re[t * kExPitch + m2] = u.x;
    ex_im[t * kExPitch + m2] = u.y;
  }
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1)
    v[k1] = make_float2(ex_re[k1 * kExPitch + t], ex_im[k1 * kExPitch + t]);
  dft16<SIGN>(v);
}

// Same, for callers whose exchange planes alias the LDS buffer the inputs were just
// read from: a barrier before the first exchange write and after the last read.
template <int SIGN>
__device__ __forceinline__ void fft256_16t_aliased(cf v[16], const cf tw[16], float* ex_re,
                                                   float* ex_im, int t) {
  __syncthreads();
  fft256_16t<SIGN>(v, tw, ex_re, ex_im, t);
  __syncthreads();
}

// Same FFT256 with its twiddles W256^(t m2) read from an LDS table (tw_t[16 m2]) instead of
// 32 registers: for callers that need the registers (k_fft_colsq).
template <int SIGN>
__device__ __forceinline__ void fft256_16t_ldstw(cf v[16], const cf* tw_t, float* ex_re, float* ex_im, int t) {
  dft16<SIGN>(v);
#pragma unroll
  for (int m2 = 0; m2 < 16; ++m2) {
    cf u = cmul(v[m2], tw_t[16 * m2]);
    ex_re[t * kExPitch + m2] = u.x;
    ex_im[t * kExPitch + m2] = u.y;
  }
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1)
    v[k1] = make_float2(ex_re[k1 * kExPitch + t], ex_im[k1 * kExPitch + t]);
  dft16<SIGN>(v);
}

// ---------------------------------------------------------------------------
// per-channel sum (fp64 accumulate).  The workgroups of a channel leave their partial sums
// in a scratch row and a second, tiny launch adds them up in index order, so the result does
// not depend on the order the workgroups ran in (repeat executes are bit-identical by
// construction; the launch boundary is the only synchronisation: a device-scope fence inside
// the kernel costs a write-back of the XCD's L2, 0.4 ms beside 50 GB of fresh results).
// sums: [C] results, then [C][kSumParts] partials.  grid (parts, C)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_channel_sum(const float* __restrict__ x, int64_t n,
                                                     double* __restrict__ sums) {
  const int c = blockIdx.y, n_ch = gridDim.y;
  const float* xc = x + (int64_t)c * n;
  // 16-byte loads over the aligned middle of the row, four independent fp64 accumulators
  const int64_t head = std::min<int64_t>(n, (4 - (((uintptr_t)xc >> 2) & 3)) & 3);
  const int64_t nv = (n - head) >> 2;
  const float4* xv = reinterpret_cast<const float4*>(xc + head);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < nv; i += 4 * stride) {      // four 16-byte loads in flight per thread
    const float4 u = xv[i], v = xv[i + stride], w = xv[i + 2 * stride], z = xv[i + 3 * stride];
    a0 += ((double)u.x + (double)v.x) + ((double)w.x + (double)z.x);
    a1 += ((double)u.y + (double)v.y) + ((double)w.y + (double)z.y);
    a2 += ((double)u.z + (double)v.z) + ((double)w.z + (double)z.z);
    a3 += ((double)u.w + (double)v.w) + ((double)w.w + (double)z.w);
  }
  for (; i < nv; i += stride) {
    const float4 u = xv[i];
    a0 += (double)u.x; a1 += (double)u.y; a2 += (double)u.z; a3 += (double)u.w;
  }
  double acc = (a0 + a1) + (a2 + a3);
  if (blockIdx.x == 0) {   // the unaligned ends of the row
    const int64_t tail0 = head + 4 * nv;
    if ((int64_t)threadIdx.x < head) acc += (double)xc[threadIdx.x];
    if (tail0 + (int64_t)threadIdx.x < n) acc += (double)xc[tail0 + threadIdx.x];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    sums[n_ch + (int64_t)c * kSumParts + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// sums[c] = partials[c][0] + partials[c][1] + ...  in index order.  grid (ceil(C / 64)), block 64
__global__ void __launch_bounds__(64) k_channel_sum_final(double* __restrict__ sums, int n_ch, int parts) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n_ch) return;
  const double* const partials = sums + n_ch + (int64_t)c * kSumParts;
  double total = 0.0;
  for (int q = 0; q < parts; ++q) total += partials[q];
  sums[c] = total;
}

// ---------------------------------------------------------------------------
// filter bank: H[s][k] = G_s(theta) exp(-i theta d),  theta = 2 pi k / (B R_s), where G_s
// is the exact (real) response of the L-tap kernel the reference convolves with
// (morse_exact.h; morseutils.py:117-149 + convolution.py:68-87) and d the half-sample
// delay of even L.  gain[s][k] = G_s(theta), signed.
// grid (S), block (B)
// ---------------------------------------------------------------------------
__global__ void k_build_bank(cf* __restrict__ bank, float* __restrict__ gain,
                             const BankScale* __restrict__ sc, const double* __restrict__ amps,
                             int B) {
  const int s = blockIdx.x;
  const int k = threadIdx.x;
  const BankScale p = sc[s];
  cf h = make_float2(0.f, 0.f);
  float g = 0.f;
  if (p.spectral) {
    const int64_t b = (int64_t)B * p.decimation;
    const int kk = k - p.band_shift;         // bins are counted from the bottom of the level's band
    const double gd = exact_gain(amps + p.amp_offset, p.bin_lo, p.n_bins, p.length, kk, b);
    double sn, cs;
    sincospi(-2.0 * (double)kk / (double)b * p.half_delay, &sn, &cs);
    h = make_float2((float)(gd * cs), (float)(gd * sn));
    g = (float)gd;
  }
  bank[(int64_t)s * B + k] = h;
  gain[(int64_t)s * B + k] = g;
}

// gain[s][k] = G_s = H_s[k] exp(+i theta d) (real) from a bank that arrived by broadcast:
// the production synthesis kernel multiplies by the real gain and folds the half-sample
// phase of even-length kernels into its persistent operand.
// grid (S), block (256)
__global__ void k_bank_gain(const cf* __restrict__ bank, float* __restrict__ gain,
                            const BankScale* __restrict__ sc) {
  const int s = blockIdx.x, k = threadIdx.x;
  const BankScale p = sc[s];
  const cf h = bank[(int64_t)s * 256 + k];
  float sn, cs;
  sincospif((float)(2.0 * (double)(k - p.band_shift) / (256.0 * (double)p.decimation) * p.half_delay), &sn, &cs);
  gain[(int64_t)s * 256 + k] = h.x * cs - h.y * sn;
}

// Which of the sixteen first-pass inputs of the synthesis (input j = bins 16 j .. 16 j + 15 of
// the scale's 256-bin block response) are negligible above the band: every bin from 16 j_hi up
// is below tol = the plan's band tolerance times the peak gain -- the same tolerance that lets
// the decimation drop everything from theta_hi up.  16 - j_hi goes into the top byte of the
// scale's entry in its level's list.  (Nothing is skipped below the band: the L-tap
// truncation's side lobes stay near 1e-8 of the peak down to zero frequency, and recordings
// carry most of their power there.)
// The kernel also lays the level's gains out the way k_synth7 parks them in LDS -- row =
// position in the level lists, lane t's sixteen gains (bins t + 16 j) side by side -- so that a
// workgroup's refill is one 16-byte load per thread at an address it knows without reading the
// scale list first (gain_lv: n_listed + 8 rows of 256 floats).  prune = 0: windows left whole.
// grid (n_listed), block (256)
__global__ void k_scale_windows(const float* __restrict__ gain, int32_t* __restrict__ scale_list, float tol,
                                float* __restrict__ gain_lv, int prune) {
  __shared__ float red[4];
  __shared__ int last;
  const int k = threadIdx.x;
  const int s = scale_list[blockIdx.x] & kScaleIndexMask;
  const float g_signed = gain[(int64_t)s * 256 + k];
  gain_lv[(int64_t)blockIdx.x * 256 + (k & 15) * 16 + (k >> 4)] = g_signed;
  const float g = fabsf(g_signed);
  float m = g;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((k & 63) == 0) red[k >> 6] = m;
  if (k == 0) last = -1;
  __syncthreads();
  const float peak = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if (g > tol * peak) atomicMax(&last, k);
  __syncthreads();
  if (k == 0) {
    const int j_hi = prune ? min(16, max(9, (last + 16) >> 4)) : 16;   // ceil((last + 1) / 16), inputs 0..8 always kept
    scale_list[blockIdx.x] = s | ((16 - j_hi) << 24);
  }
}

// ---------------------------------------------------------------------------
// literal kernels for direct scales: psi[n] = (1/L) sum_j A_j e^{i pi j (L+1)/L} e^{2 pi i j n / L}
// over the scale's kept spectrum samples A_j (planner.h: amps)    (morseutils.py:147-149)
// grid (n_direct, ceil(Lmax/256))
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_direct(cf* __restrict__ psi, const DirectScale* __restrict__ sc,
                                                      const double* __restrict__ amps, cf* __restrict__ tail,
                                                      cf* __restrict__ psi_literal) {
  // The taps k_direct uses are the RUNNING SUMS Psi[j] = psi[0] + .. + psi[j], j < L - 1, applied to the first
  // difference of the signal (summation by parts, exact:
  //   sum_j x[q - j] psi[j] = Psi[L-1] x[q - L + 1] + sum_{j < L-1} Psi[j] (x[q - j] - x[q - j - 1]) ),
  // so that float32 products are made with the increments of the recording, not with its level: a slow
  // background 100 x a quiet band no longer costs that band its low bits (round 4).  Tap L - 1 is zero;
  // tail[scale] = Psi[L-1], the kernel's response at zero frequency, multiplies x itself.
  __shared__ double sre[256], sim[256];
  const DirectScale p = sc[blockIdx.x];
  const int64_t L = p.length;
  double carry_re = 0.0, carry_im = 0.0;
  for (int64_t n0 = 0; n0 < L; n0 += 256) {
    const int64_t n = n0 + threadIdx.x;
    double re = 0.0, im = 0.0;
    if (n < L) {
      for (int32_t i = 0; i < p.n_bins; ++i) {
        const int64_t k = p.bin_lo + i;
        // phase = pi k (L+1)/L + 2 pi k n / L, reduced exactly: (k (L+1 + 2n)) mod 2L over L
        const int64_t q = (k * ((L + 1 + 2 * n) % (2 * L))) % (2 * L);
        double sn, cs;
        sincospi((double)q / (double)L, &sn, &cs);
        const double a = amps[p.amp_offset + i];
        re += a * cs;
        im += a * sn;
      }
      re /= (double)L;
      im /= (double)L;
    }
    if (n < L) psi_literal[p.offset + p.front + n] = make_float2((float)re, (float)im);   // (gcwt_direct_kernel: tests)
    sre[threadIdx.x] = re;
    sim[threadIdx.x] = im;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {               // inclusive scan
      const double ar = threadIdx.x >= off ? sre[threadIdx.x - off] : 0.0;
      const double ai = threadIdx.x >= off ? sim[threadIdx.x - off] : 0.0;
      __syncthreads();
      sre[threadIdx.x] += ar;
      sim[threadIdx.x] += ai;
      __syncthreads();
    }
    const double pr = carry_re + sre[threadIdx.x], pi = carry_im + sim[threadIdx.x];
    if (n < L - 1) psi[p.offset + p.front + n] = make_float2((float)pr, (float)pi);
    else if (n == L - 1) { psi[p.offset + p.front + n] = make_float2(0.f, 0.f); tail[blockIdx.x] = make_float2((float)pr, (float)pi); }
    carry_re += sre[255];
    carry_im += sim[255];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// In-LDS radix-2 decimation-in-frequency FFT of `nfft` independent transforms
// of length `len`; element i of transform f lives at buf[f*fstride + i*istride].
// Result k ends up at element bitrev(k).  All 256 threads take part.
// tw4096[j] = exp(-2 pi i j / 4096), j < 2048 (conjugated for SIGN > 0).
// ---------------------------------------------------------------------------
template <int SIGN>
__device__ __forceinline__ void lds_fft_radix2(cf* buf, int len, int log2len, int nfft,
                                               int fstride, int istride,
                                               const cf* __restrict__ tw4096) {
  const int half_total = nfft * (len >> 1);
  const bool f_fast = fstride < istride;  // column tiles: neighbouring threads take neighbouring f
  for (int h = len >> 1, st = 0; h >= 1; h >>= 1, ++st) {
    for (int b = threadIdx.x; b < half_total; b += 256) {
      int f, j;
      if (f_fast) { f = b % nfft; j = b / nfft; } else { f = b / (len >> 1); j = b % (len >> 1); }
      const int pos = j & (h - 1);
      const int i0 = ((j - pos) << 1) + pos;
      cf* p0 = buf + f * fstride + i0 * istride;
      cf* p1 = p0 + h * istride;
      const cf a = *p0, c = *p1;
      cf w = tw4096[(pos << st) * (kRowLenDev >> log2len)];
      if (SIGN > 0) w.y = -w.y;
      *p0 = cadd(a, c);
      *p1 = cmul(csub(a, c), w);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int v, int bits) {
  return bits ? (int)(__brev((unsigned)v) >> (32 - bits)) : 0;
}

// ---------------------------------------------------------------------------
// Column pass: for a tile of 16 adjacent columns, FFT over the row index.
//   in  element (i, col) at in[i*ld + col],  i < len (= P1), col < ld
//   out element (k, col) at out[k*ld + col], multiplied by
//       exp(SIGN 2 pi i k col / tw_n) when tw_n > 0.
// REAL_IN: input is the float32 signal: element (i, col) is sample i*ld + col of
// the epoch, minus the channel mean, zero beyond n_valid.  (transforms.py:142-143)
// grid (ld/16, C), dynamic LDS len*16*8 bytes
// ---------------------------------------------------------------------------
template <int SIGN, bool REAL_IN>
__global__ void __launch_bounds__(256) k_fft_cols(const void* __restrict__ in_, cf* __restrict__ out,
                                                  int len, int log2len, int ld, int64_t in_cstride,
                                                  int64_t out_cstride, int64_t tw_n,
                                                  const cf* __restrict__ tw4096,
                                                  const double* __restrict__ sums, double inv_n,
                                                  const SegIn segs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* buf = reinterpret_cast<cf*>(smem);
  const int c = blockIdx.y;               // workspace slot: segment * n_channels + channel
  const int col0 = blockIdx.x * 16;
  const int total = len * 16;
  if (REAL_IN) {
    const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
    const int64_t n_valid = segs.n_valid[g], n_lead = segs.n_lead[g];
    const float* x = reinterpret_cast<const float*>(in_) + (int64_t)ch * in_cstride + segs.x_off[g];
    const double mean = sums[ch] * inv_n;   // subtracted in fp64: a mean 1e3 x the signal's spread must not cost its low bits
    for (int e = threadIdx.x; e < total; e += 256) {
      const int i = e >> 4, cc = e & 15;
      const int64_t n = (int64_t)i * ld + col0 + cc;
      buf[e] = make_float2(n >= n_lead && n < n_valid ? (float)((double)x[n] - mean) : 0.f, 0.f);
    }
  } else {
    const cf* x = reinterpret_cast<const cf*>(in_) + (int64_t)c * in_cstride;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int i = e >> 4, cc = e & 15;
      buf[e] = x[(int64_t)i * ld + col0 + cc];
    }
  }
  __syncthreads();
  if (len > 1) lds_fft_radix2<SIGN>(buf, len, log2len, 16, 1, 16, tw4096);
  cf* o = out + (int64_t)c * out_cstride;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int k = e >> 4, cc = e & 15;
    cf v = buf[bitrev(k, log2len) * 16 + cc];
    if (tw_n > 0) {
      const int64_t q = ((int64_t)k * (col0 + cc)) % tw_n;
      float sn, cs;
      sincospif(2.0f * (float)q / (float)tw_n, &sn, &cs);
      v = cmul(v, make_float2(cs, SIGN > 0 ? sn : -sn));
    }
    o[(int64_t)k * ld + col0 + cc] = v;
  }
}

// ---------------------------------------------------------------------------
// Row pass: contiguous FFTs of length len <= 4096; one workgroup transforms
// 4096/len rows.  Row r reads in[r*in_ld .. +len) and writes out[r*out_ld .. +len),
// multiplied by exp(SIGN 2 pi i r k / tw_n) when tw_n > 0, times `scale`.
// grid (n_rows / (4096/len), C), LDS 32 KiB
// ---------------------------------------------------------------------------
template <int SIGN>
__global__ void __launch_bounds__(256) k_fft_rows(const cf* __restrict__ in, cf* __restrict__ out,
                                                  int len, int log2len, int64_t in_ld,
                                                  int64_t out_ld, int64_t in_cstride,
                                                  int64_t out_cstride, int64_t tw_n,
                                                  const cf* __restrict__ tw4096, float scale,
                                                  int n_rows, const RowTaper tp) {
  __shared__ __attribute__((aligned(16))) cf buf[kRowLenDev];
  const int c = blockIdx.y;
  const int rows = kRowLenDev >> log2len;
  const int row0 = blockIdx.x * rows;
  const cf* x = in + (int64_t)c * in_cstride;
  for (int e = threadIdx.x; e < kRowLenDev; e += 256) {
    const int r = e >> log2len, i = e & (len - 1);
    cf v = row0 + r < n_rows ? x[(int64_t)(row0 + r) * in_ld + i] : make_float2(0.f, 0.f);
    if (tp.p1) {
      const float f = row_taper(tp, row0 + r, i);
      v = make_float2(v.x * f, v.y * f);
    }
    buf[e] = v;
  }
  __syncthreads();
  lds_fft_radix2<SIGN>(buf, len, log2len, rows, len, 1, tw4096);
  cf* o = out + (int64_t)c * out_cstride;
  for (int e = threadIdx.x; e < kRowLenDev; e += 256) {
    const int r = e >> log2len, k = e & (len - 1);
    if (row0 + r >= n_rows) continue;
    cf v = buf[(r << log2len) + bitrev(k, log2len)];
    if (tw_n > 0) {
      const int64_t q = ((int64_t)(row0 + r) * k) % tw_n;
      float sn, cs;
      sincospif(2.0f * (float)q / (float)tw_n, &sn, &cs);
      v = cmul(v, make_float2(cs, SIGN > 0 ? sn : -sn));
    }
    o[(int64_t)(row0 + r) * out_ld + k] = make_float2(v.x * scale, v.y * scale);
  }
}

// ---------------------------------------------------------------------------
// Fast paths of the big FFT: register radix-16 instead of radix-2 LDS stages.
// ---------------------------------------------------------------------------
template <int SIGN>
__device__ __forceinline__ void dft2(cf& a, cf& b) {
  const cf s = cadd(a, b), d = csub(a, b);
  a = s;
  b = d;
}

template <int SIGN>
__device__ __forceinline__ void dft8(cf v[8]) {
  const float h = 0.70710678118654752f, sg = (float)SIGN;
  dft4<SIGN>(v[0], v[2], v[4], v[6]);      // E[0..3] in v[0], v[2], v[4], v[6]
  dft4<SIGN>(v[1], v[3], v[5], v[7]);      // O[0..3] in v[1], v[3], v[5], v[7]
  const cf o1 = cmul(v[3], make_float2(h, sg * h));
  const cf o2 = mul_si<SIGN>(v[5]);
  const cf o3 = cmul(v[7], make_float2(-h, sg * h));
  const cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

template <int SIGN, int Q>
__device__ __forceinline__ void dft_small(cf v[Q]) {
  if (Q == 2) dft2<SIGN>(v[0], v[1]);
  else if (Q == 4) dft4<SIGN>(v[0], v[1], v[2], v[3]);
  else if (Q == 8) dft8<SIGN>(v);
  else if (Q == 16) dft16<SIGN>(v);
}

// exp(SIGN 2 pi i idx/4096) from the half table exp(-2 pi i j/4096), j < 2048
template <int SIGN>
__device__ __forceinline__ cf tw4096_at(const cf* __restrict__ tw4096, int idx) {
  cf w = tw4096[idx & 2047];
  if (idx & 2048) w = make_float2(-w.x, -w.y);
  if (SIGN > 0) w.y = -w.y;
  return w;
}

__device__ __forceinline__ cf unit_phase(int64_t num, int64_t den, int sign) {
  float sn, cs;
  // every length here is a power of two: the remainder is a mask (the general form stays for safety)
  const int64_t r = (den & (den - 1)) == 0 ? (num & (den - 1)) : num % den;
  sincospif(2.0f * (float)r / (float)den, &sn, &cs);
  return make_float2(cs, sign > 0 ? sn : -sn);
}

__device__ __forceinline__ int pad32(int i) { return i + (i >> 5); }
// element buffer of the 4096-point transforms' last exchange: thread T's sixteen values 18 elements apart -- 36 dwords,
// = 4 (mod 32), so the 16-byte reads of eight neighbouring lanes cover the 32 banks once (at a pitch of 16 + T / 2
// pairs of lanes met on the same banks: 17 % of the LDS time of k_bc_scales, profiles/r04_pmc_k_bc_scales.json)
constexpr int kElemPitch = 18;
static_assert(256 * kElemPitch <= 16 * kExColD, "the element buffer aliases the exchange planes");


// Rows of length Q = 256 q, q = 1 << LQ: FFT256 over the stride-q subsequences,
// twiddle W_Q^(kb a), DFT_q over a; output index kb + 256 ka is contiguous in kb.
template <int SIGN, int LQ>
__device__ __forceinline__ void rows_fast_body(const cf* __restrict__ x, cf* __restrict__ o,
                                               cf* buf, float* ex_re, float* ex_im, int64_t in_ld,
                                               int64_t out_ld, int64_t tw_n,
                                               const cf* __restrict__ tw4096,
                                               const cf* __restrict__ tw256, float scale, int row0,
                                               int n_rows, int out_len, int mirror, const RowTaper& tp) {
  constexpr int q = 1 << LQ, Q = 256 * q, rows = 16 >> LQ;
  const int tid = threadIdx.x;
  // Lanes run over the 16 interleaved subsequences (row, a) so that each load instruction
  // reads runs of q consecutive elements per row, straight into the registers of the thread
  // that transforms them; for q = 1 lanes run along the row instead.
  const int s = LQ == 0 ? tid >> 4 : tid & 15, t = LQ == 0 ? tid & 15 : tid >> 4;
  constexpr int col_stride = LQ == 0 ? kExCol : kExColD;
  const int rr = s >> LQ, a = s & (q - 1);
  cf tw[16], v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cf w = tw256[(t * j) & 255];              // table holds exp(+2 pi i q/256)
    if (SIGN < 0) w.y = -w.y;
    tw[j] = w;
  }
  {
    const bool live = row0 + rr < n_rows;
    const cf* xp = x + (int64_t)min(row0 + rr, n_rows - 1) * in_ld + q * t + a;   // clamped row
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const cf u = xp[q * 16 * j];
      v[j] = live ? u : make_float2(0.f, 0.f);
    }
    if (tp.p1 && ((float)(row0 + rr + tp.p1 * (q * t + a)) - tp.k0) * tp.inv_width < 1.f) {
      // (the thread's lowest bin lies inside the cut: its higher ones may too)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float f = row_taper(tp, row0 + rr, q * t + a + q * 16 * j);
        v[j] = make_float2(v[j].x * f, v[j].y * f);
      }
    }
  }
  fft256_16t<SIGN>(v, tw, ex_re + s * col_stride, ex_im + s * col_stride, t);
  if (q == 1) {
    const int row = row0 + rr;
    if (row < n_rows) {
      cf w = make_float2(1.f, 0.f), st = w;
      if (tw_n > 0) {
        w = unit_phase((int64_t)row * t, tw_n, SIGN);
        st = unit_phase((int64_t)row * 16, tw_n, SIGN);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        cf val = v[j];
        if (tw_n > 0) { val = cmul(val, w); w = cmul(w, st); }
        if (t + 16 * j < out_len)
          o[(int64_t)row * out_ld + t + 16 * j] = make_float2(val.x * scale, val.y * scale);
      }
    }
    return;
  }
  __syncthreads();   // the element buffer below aliases the exchange planes
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kb = t + 16 * j;
    const cf w = tw4096_at<SIGN>(tw4096, kb * a * (16 >> LQ));
    buf[pad32(rr * Q + q * kb + a)] = cmul(v[j], w);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < rows; ++i) {
    const int item = tid + 256 * i;
    const int r2 = item >> 8, kb = item & 255;
    const int row = row0 + r2;
    cf u[q];
#pragma unroll
    for (int aa = 0; aa < q; ++aa) u[aa] = buf[pad32(r2 * Q + q * kb + aa)];
    dft_small<SIGN, q>(u);
    if (row < n_rows) {
      cf w = make_float2(1.f, 0.f), st = w;
      if (tw_n > 0) {
        w = unit_phase((int64_t)row * kb, tw_n, SIGN);
        st = unit_phase((int64_t)row * 256, tw_n, SIGN);
      }
#pragma unroll
      for (int ka = 0; ka < q; ++ka) {
        cf val = u[ka];
        if (tw_n > 0) { val = cmul(val, w); w = cmul(w, st); }
        const int idx = kb + 256 * ka;
        if (mirror == 0) {
          if (256 * ka < out_len)   // out_len is a multiple of 256 (or the whole row)
            o[(int64_t)row * out_ld + idx] = make_float2(val.x * scale, val.y * scale);
        } else if (idx < Q / 2) {
          o[(int64_t)row * out_ld + idx] = make_float2(val.x * scale, val.y * scale);
        } else if (row > 0 && 2 * row < mirror) {
          // spectrum of a real signal, k1-major: X[(P1 - k1) + P1 k2] = conj(X[k1 + P1 (Q-1-k2)]),
          // so the upper half of row k1 is the lower half of row P1 - k1, reversed
          o[(int64_t)(mirror - row) * out_ld + (Q - 1 - idx)] = make_float2(val.x * scale, -val.y * scale);
        }
      }
    }
  }
}

template <int SIGN>
__global__ void __launch_bounds__(256, 4) k_fft_rows_fast(const cf* __restrict__ in, cf* __restrict__ out,
                                                       int lq, int64_t in_ld, int64_t out_ld,
                                                       int64_t in_cstride, int64_t out_cstride,
                                                       int64_t tw_n, const cf* __restrict__ tw4096,
                                                       const cf* __restrict__ tw256, float scale,
                                                       int n_rows, int out_len, int mirror, const RowTaper tp) {
  // the exchange planes (2 x 16 x 290 floats) alias the 4096(+128 pad)-element buffer of
  // the second stage
  __shared__ __attribute__((aligned(16))) cf buf[16 * kExColD];
  float* const ex_re = reinterpret_cast<float*>(buf);
  float* const ex_im = ex_re + 16 * kExColD;
  const cf* x = in + (int64_t)blockIdx.y * in_cstride;
  cf* o = out + (int64_t)blockIdx.y * out_cstride;
  const int row0 = blockIdx.x * (16 >> lq);
  switch (lq) {
    case 0: rows_fast_body<SIGN, 0>(x, o, buf, ex_re, ex_im, in_ld, out_ld, tw_n, tw4096, tw256, scale, row0, n_rows, out_len, mirror, tp); break;
    case 1: rows_fast_body<SIGN, 1>(x, o, buf, ex_re, ex_im, in_ld, out_ld, tw_n, tw4096, tw256, scale, row0, n_rows, out_len, mirror, tp); break;
    case 2: rows_fast_body<SIGN, 2>(x, o, buf, ex_re, ex_im, in_ld, out_ld, tw_n, tw4096, tw256, scale, row0, n_rows, out_len, mirror, tp); break;
    case 3: rows_fast_body<SIGN, 3>(x, o, buf, ex_re, ex_im, in_ld, out_ld, tw_n, tw4096, tw256, scale, row0, n_rows, out_len, mirror, tp); break;
    default: rows_fast_body<SIGN, 4>(x, o, buf, ex_re, ex_im, in_ld, out_ld, tw_n, tw4096, tw256, scale, row0, n_rows, out_len, mirror, tp); break;
  }
}

// Column pass for len = 256: 16 columns per workgroup, one FFT256 per column by 16 threads.
// Lanes run over the columns (16 consecutive complex = 128 bytes per row), so the inputs go
// from global memory straight into the registers of the thread that transforms them and
// the outputs straight back: the only LDS traffic is the FFT's own 16 x 16 exchange.  The
// exchange planes of neighbouring columns are 290 floats apart: with lanes = columns that
// keeps the 32 lanes of a half-wave on different banks (2 s + t).

template <int SIGN, bool REAL_IN>
__global__ void __launch_bounds__(256, 4) k_fft_cols256(const void* __restrict__ in_, cf* __restrict__ out,
                                                     int ld, int64_t in_cstride, int64_t out_cstride,
                                                     int64_t tw_n, const cf* __restrict__ tw256,
                                                     const double* __restrict__ sums, double inv_n,
                                                     const SegIn segs, int rows_out) {
  __shared__ float ex_re[16 * kExColD];
  __shared__ float ex_im[16 * kExColD];
  const int c = blockIdx.y, col0 = blockIdx.x * 16, tid = threadIdx.x;   // c: workspace slot
  const int s = tid & 15, t = tid >> 4;
  cf tw[16], v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cf w = tw256[(t * j) & 255];
    if (SIGN < 0) w.y = -w.y;
    tw[j] = w;
  }
  if (REAL_IN) {
    const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
    const int64_t n_valid = segs.n_valid[g], n_lead = segs.n_lead[g];
    const float* x = reinterpret_cast<const float*>(in_) + (int64_t)ch * in_cstride + segs.x_off[g];
    const double mean = sums[ch] * inv_n;   // subtracted in fp64: a mean 1e3 x the signal's spread must not cost its low bits
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t n = (int64_t)(t + 16 * j) * ld + col0 + s;
      const float a = x[min(max(n, n_lead), n_valid - 1)];   // clamped: no branch around the load
      v[j] = make_float2(n >= n_lead && n < n_valid ? (float)((double)a - mean) : 0.f, 0.f);
    }
  } else {
    const cf* x = reinterpret_cast<const cf*>(in_) + (int64_t)c * in_cstride + col0 + s;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = x[(int64_t)(t + 16 * j) * ld];
  }
  fft256_16t<SIGN>(v, tw, ex_re + s * kExColD, ex_im + s * kExColD, t);
  cf w = make_float2(1.f, 0.f), st = w;
  if (tw_n > 0) {
    w = unit_phase((int64_t)(col0 + s) * t, tw_n, SIGN);
    st = unit_phase((int64_t)(col0 + s) * 16, tw_n, SIGN);
  }
  // real input: rows k and 256 - k are conjugates, the caller may ask for 0 .. 128 only
  cf* o = out + (int64_t)c * out_cstride + col0 + s;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cf val = v[j];
    if (tw_n > 0) { val = cmul(val, w); w = cmul(w, st); }
    if (t + 16 * j < rows_out) o[(int64_t)(t + 16 * j) * ld] = val;
  }
}

// Column pass for len = 256 q, q = 2 or 4 (FFT lengths 2^21 and 2^22: long recordings, time
// blocks): the same 16-column tile and lanes-over-columns loads as k_fft_cols256; the rows
// are taken as q interleaved subsequences, each through the register FFT256, times
// W_len^(a kb); a thread ends up with all q values of its own 16 (kb, column) pairs -- kept
// in registers, one set of four in LDS (parking the first q-1 in LDS took 98 KB for q = 4: one
// workgroup, i.e. one wave per SIMD, per CU, and 1.26 ms per 24 x 2^22 launch) -- and
// finishes them with a DFT_q.
// grid (ld/16, slots), 39 KB (q = 2) or 71 KB (q = 4) of LDS, two workgroups per CU (256
// VGPRs; q = 4 still spills about 30 of its parked values to scratch, each stored and read
// once): 0.5 ms per 24 x 2^22 launch
template <int SIGN, bool REAL_IN, int LQ>
__global__ void __launch_bounds__(256, 2) k_fft_colsq(const void* __restrict__ in_, cf* __restrict__ out,
                                                   int ld, int64_t in_cstride, int64_t out_cstride,
                                                   int64_t tw_n, const cf* __restrict__ tw4096,
                                                   const cf* __restrict__ tw256,
                                                   const double* __restrict__ sums, double inv_n,
                                                   const SegIn segs, int rows_out) {
  constexpr int q = 1 << LQ, len = 256 * q;
  __shared__ float ex_re[16 * kExColD];
  __shared__ float ex_im[16 * kExColD];
  const int c = blockIdx.y, col0 = blockIdx.x * 16, tid = threadIdx.x;   // c: workspace slot
  const int s = tid & 15, t = tid >> 4;
  __shared__ cf twl[256];                 // W256^(+-t j) at [j][t]: the FFT256's twiddles, read as broadcasts
  {
    cf w = tw256[((tid & 15) * (tid >> 4)) & 255];
    if (SIGN < 0) w.y = -w.y;
    twl[tid] = w;
  }
  // q = 4: the first subsequence's values wait in LDS (32 KB of this thread's own words: no
  // barrier), the next two in registers; q = 2: the one in registers
  constexpr int kInLds = q == 4 ? 1 : 0, kInReg = q - 1 - kInLds;
  __shared__ cf park[kInLds ? 16 * 256 : 1];
  cf v[16], u[kInReg * 16];
#pragma unroll
  for (int a = 0; a < q; ++a) {
    if (REAL_IN) {
      const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
      const int64_t n_valid = segs.n_valid[g], n_lead = segs.n_lead[g];
      const float* x = reinterpret_cast<const float*>(in_) + (int64_t)ch * in_cstride + segs.x_off[g];
      const double mean = sums[ch] * inv_n;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int64_t n = (int64_t)(q * (t + 16 * j) + a) * ld + col0 + s;
        const float xv = x[min(max(n, n_lead), n_valid - 1)];   // clamped: no branch around the load
        v[j] = make_float2(n >= n_lead && n < n_valid ? (float)((double)xv - mean) : 0.f, 0.f);
      }
    } else {
      const cf* x = reinterpret_cast<const cf*>(in_) + (int64_t)c * in_cstride + col0 + s;
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = x[(int64_t)(q * (t + 16 * j) + a) * ld];
    }
    __syncthreads();                      // twiddle table written / previous exchange read
    fft256_16t_ldstw<SIGN>(v, twl + t, ex_re + s * kExColD, ex_im + s * kExColD, t);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int kb = t + 16 * j;
      const cf val = cmul(v[j], tw4096_at<SIGN>(tw4096, a * kb * (kRowLenDev / len)));
      if (a < kInLds) park[j * 256 + tid] = val;
      else if (a < q - 1) u[(a < q - 1 ? a - kInLds : 0) * 16 + j] = val;
      else v[j] = val;
    }
  }
  cf* o = out + (int64_t)c * out_cstride + col0 + s;
  // output twiddle W_P^(col k), k = t + 16 j + 256 ka: three sincos per thread and the same
  // 16-step recurrence over j as k_fft_cols256, three more steps over ka
  cf stq = make_float2(1.f, 0.f), st16 = stq, wj = stq;
  if (tw_n > 0) {
    stq = unit_phase((int64_t)(col0 + s) * 256, tw_n, SIGN);
    st16 = unit_phase((int64_t)(col0 + s) * 16, tw_n, SIGN);
    wj = unit_phase((int64_t)(col0 + s) * t, tw_n, SIGN);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kb = t + 16 * j;
    cf w[q];
#pragma unroll
    for (int a = 0; a < q - 1; ++a) w[a] = a < kInLds ? park[j * 256 + tid] : u[(a < kInLds ? 0 : a - kInLds) * 16 + j];
    w[q - 1] = v[j];
    dft_small<SIGN, q>(w);
    cf ph = wj;
#pragma unroll
    for (int ka = 0; ka < q; ++ka) {
      const int k = kb + 256 * ka;
      if (k < rows_out) {
        cf val = w[ka];
        if (tw_n > 0) val = cmul(val, ph);
        o[(int64_t)k * ld] = val;
      }
      if (tw_n > 0) ph = cmul(ph, stq);
    }
    if (tw_n > 0) wj = cmul(wj, st16);
  }
}

// ---- the full-band path's two fused passes (exact.hip describes the path) -------------------------------
// Pass 1: Z_s = IFFT rows of (X * H_s) for up to four scales: the 4096-point inverse row pass of
// rows_fast_body<+1, 4> with the product folded into its loads; a row of X is read once for all of them.
// grid (slots * P1): workgroup id -> (row % group, slot, row / group), so that the workgroups that run together
// share `group` rows of H (32 KB each, L2) without all of them reading the same offset of 8 MB-strided slots
__global__ void __launch_bounds__(256, 2) k_fullband_rows(const cf* __restrict__ in, const FullbandSet set,
                                                       int64_t in_cstride, int64_t out_cstride, int64_t tw_n,
                                                       const cf* __restrict__ twt,
                                                       const cf* __restrict__ tw256, int n_slots, int group) {
  __shared__ __attribute__((aligned(16))) cf buf[16 * kExColD];
  float* const ex_re = reinterpret_cast<float*>(buf) + (threadIdx.x & 15) * kExColD;
  float* const ex_im = ex_re + 16 * kExColD;
  __shared__ v2f twl[256];
  const int tid = threadIdx.x, a = tid & 15, t = tid >> 4;
  const int bid = blockIdx.x, per = group * n_slots;
  const int slot = (bid % per) / group, row = bid % group + group * (bid / per);
  {
    const cf w = tw256[(a * t) & 255];
    twl[tid] = v2f{w.x, w.y};
  }
  const int64_t at = (int64_t)row * kRowLenDev + 16 * t + a;
  v2f xv[16], v[16];
  {
    const v2f* xp = reinterpret_cast<const v2f*>(in) + (int64_t)slot * in_cstride + at;
#pragma unroll
    for (int j = 0; j < 16; ++j) xv[j] = xp[256 * j];
  }
  v2f w0 = {1.f, 0.f}, st = w0;
  if (tw_n > 0) {
    const cf p0 = unit_phase((int64_t)row * tid, tw_n, 1), p1 = unit_phase((int64_t)row * 256, tw_n, 1);
    w0 = v2f{p0.x, p0.y};
    st = v2f{p1.x, p1.y};
  }
  v2f* const bufv = reinterpret_cast<v2f*>(buf);
  const v2f* const tws = reinterpret_cast<const v2f*>(twt) + tid;
  v2f twa[4], twb[4];                   // eight of the middle twiddles, the sixteen are their products (k_bc_scales)
#pragma unroll
  for (int m = 0; m < 4; ++m) twa[m] = tws[256 * (4 * m)];
#pragma unroll
  for (int n = 0; n < 4; ++n) twb[n] = reinterpret_cast<const v2f*>(twt)[256 * n + a];
  // the three radix-16 layers are synth_math.h's packed idft16v, as in k_bc_scales (output k in register dft16_pos(k))
  // the next scale's row of H is on its way while this one's transform runs (the rows come from beyond L2: the
  // waves sat waiting 74 % of their cycles without it)
  v2f hc[16];
  {
    const v2f* __restrict__ hs = reinterpret_cast<const v2f*>(set.h[0]) + at;
#pragma unroll
    for (int j = 0; j < 16; ++j) hc[j] = hs[256 * j];
  }
  for (int sel = 0; sel < set.n; ++sel) {
    v2f* __restrict__ o = reinterpret_cast<v2f*>(set.z[sel]) + (int64_t)slot * out_cstride + (int64_t)row * kRowLenDev;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmulv(xv[j], hc[j]);
    if (sel + 1 < set.n) {
      const v2f* __restrict__ hs = reinterpret_cast<const v2f*>(set.h[sel + 1]) + at;
#pragma unroll
      for (int j = 0; j < 16; ++j) hc[j] = hs[256 * j];
    }
    __syncthreads();                      // twiddle table written / the buffer's last readers done
    idft16v(v);
#pragma unroll
    for (int m2 = 0; m2 < 16; ++m2) {
      const v2f u = cmulv(v[dft16_pos(m2)], twl[t + 16 * m2]);
      ex_re[t * kExPitch + m2] = u.x;
      ex_im[t * kExPitch + m2] = u.y;
    }
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = v2f{ex_re[k1 * kExPitch + t], ex_im[k1 * kExPitch + t]};
    idft16v(v);
    __syncthreads();                      // the element buffer aliases the exchange planes
#pragma unroll
    for (int j = 0; j < 16; ++j)        // W_4096^(+(t + 16 j) a) = twa[j >> 2] twb[j & 3], as in k_bc_scales
      bufv[kElemPitch * (t + 16 * j) + a] = cmulv(cmulv(v[dft16_pos(j)], twa[j >> 2]), twb[j & 3]);
    __syncthreads();
#pragma unroll
    for (int aa = 0; aa < 16; ++aa) v[aa] = bufv[kElemPitch * tid + aa];
    idft16v(v);
    v2f w = w0;
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      v2f val = v[dft16_pos(ka)];
      if (tw_n > 0) { val = cmulv(val, w); w = cmulv(w, st); }
      o[tid + 256 * ka] = val;
    }
  }
}

// P = 4 x 4096 (time blocks of 16 384 samples: kernels of 1 - 5 K taps under precision = exact, planner.h): both passes
// in one workgroup.  The four rows k1 of a slot go through k_fullband_rows' arithmetic one after the other -- product
// with the scale's response, 4096-point inverse transform, W_P^(k1 n2) -- and stay in registers; a DFT4 over k1
// finishes the column pass for the thread's sixteen n2, and the samples go to the scale's row through the sink below.
// HBM sees the slot's spectrum (128 KB, again per scale: L2 mostly) and the result: no z buffer, no column pass, no
// store pass (40 -> 11 bytes per point).  grid (slots, sets of scales)
struct Fullband4Set {
  const cf* h[kFullband4Scales];       // responses on the 16 384-point grid, k1-major like one slot of x
  int32_t scale[kFullband4Scales];     // their output rows
  int32_t n, pad;
};

// Pass 2 stores what the column transform leaves in registers: sample n = k ld + column of the slot's
// segment, cropped to the segment's window, as |.|, |.|^2 or the complex value, into the scale's row.
struct FullbandSink {
  float* out;                 // the slot's channel, the scale's row, column of segment sample 0
  int64_t w_lo, w_hi;
};

__device__ __forceinline__ FullbandSink fullband_sink(float* out, int slot, int scale, int n_scales,
                                                      int64_t row_len, const SegOut& seg, int elem) {
  const int g = slot / seg.n_channels, ch = slot - g * seg.n_channels;
  FullbandSink k;
  k.out = out + (((int64_t)ch * n_scales + scale) * row_len + seg.seg_col[g]) * elem;
  k.w_lo = seg.w_lo[g];
  k.w_hi = seg.w_hi[g];
  return k;
}

template <int MODE>
__device__ __forceinline__ void fullband_put(const FullbandSink& k, int64_t n, cf v) {
  if (n < k.w_lo || n >= k.w_hi) return;
  if (MODE == GCWT_OUT_AMPLITUDE_F32) k.out[n] = sqrtf(v.x * v.x + v.y * v.y);
  else if (MODE == GCWT_OUT_POWER_F32) k.out[n] = v.x * v.x + v.y * v.y;
  else reinterpret_cast<cf*>(k.out)[n] = v;
}

template <int MODE>
__global__ void __launch_bounds__(512, 2) k_fullband4(const cf* __restrict__ in, const Fullband4Set set,
                                                   float* __restrict__ out, int64_t in_cstride,
                                                   const cf* __restrict__ twt, const cf* __restrict__ tw256,
                                                   int n_scales, int64_t row_len, const SegOut seg) {
  // 512 threads: waves 0 - 3 take rows k1 = 0, 2, waves 4 - 7 rows 1, 3, each half with its own exchange buffer.
  // With a = z_0 +- z_2 and b = z_1 +- z_3 (W_P^(k1 n2) applied) the DFT4 over k1 is
  //   y[n2] = a+ + b+,  y[n2 + 2 . 4096] = a+ - b+,  y[n2 + 4096] = a- + i b-,  y[n2 + 3 . 4096] = a- - i b-:
  // the halves swap a- and b+ through LDS, the first stores n1 = 0, 2, the second n1 = 1, 3.
  constexpr int64_t kP = (int64_t)4 * kRowLenDev;
  __shared__ __attribute__((aligned(16))) cf bufs[2][16 * kExColD];
  __shared__ v2f twl[256];
  const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, a = tid & 15, t = tid >> 4;
  cf* const buf = bufs[half];
  float* const ex_re = reinterpret_cast<float*>(buf) + a * kExColD;
  float* const ex_im = ex_re + 16 * kExColD;
  const int slot = blockIdx.x;
  if (half == 0) {
    const cf w = tw256[(a * t) & 255];
    twl[tid] = v2f{w.x, w.y};
  }
  const int at = 16 * t + a;
  v2f* const bufv = reinterpret_cast<v2f*>(buf);
  const v2f* const tws = reinterpret_cast<const v2f*>(twt) + tid;
  v2f twa[4], twb[4];                   // the middle twiddles of the 4096-point transform (k_bc_scales)
#pragma unroll
  for (int m = 0; m < 4; ++m) twa[m] = tws[256 * (4 * m)];
#pragma unroll
  for (int n = 0; n < 4; ++n) twb[n] = reinterpret_cast<const v2f*>(twt)[256 * n + a];
  // W_P^(k1 n2), n2 = tid + 256 ka: first value and step, for this half's two rows (row 0: one)
  v2f w0[2], wst[2];
#pragma unroll
  for (int rho = 0; rho < 2; ++rho) {
    const int r = half + 2 * rho;
    const cf p0 = unit_phase((int64_t)r * tid, kP, 1), p1 = unit_phase((int64_t)r * 256, kP, 1);
    w0[rho] = v2f{p0.x, p0.y};
    wst[rho] = v2f{p1.x, p1.y};
  }
  const v2f* const xs = reinterpret_cast<const v2f*>(in) + (int64_t)slot * in_cstride + at;
  v2f* const swap_mine = reinterpret_cast<v2f*>(bufs[half]);          // what this half hands over
  const v2f* const swap_theirs = reinterpret_cast<const v2f*>(bufs[half ^ 1]);
  for (int sel = blockIdx.y; sel < set.n; sel += gridDim.y) {
    const v2f* const hs = reinterpret_cast<const v2f*>(set.h[sel]) + at;
    v2f keep[16], v[16];
#pragma nounroll
    for (int rho = 0; rho < 2; ++rho) {    // (rolled: unrolled, the second row's loads are hoisted over the first's transform)
      {
        const int64_t ro = (int64_t)(half + 2 * rho) * kRowLenDev;
        const v2f* __restrict__ xp = xs + ro;
        const v2f* __restrict__ hp = hs + ro;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = cmulv(xp[256 * j], hp[256 * j]);
      }
      __syncthreads();                    // twiddle table written / the buffers' last readers done
      idft16v(v);
#pragma unroll
      for (int m2 = 0; m2 < 16; ++m2) {
        const v2f u = cmulv(v[dft16_pos(m2)], twl[t + 16 * m2]);
        ex_re[t * kExPitch + m2] = u.x;
        ex_im[t * kExPitch + m2] = u.y;
      }
      __syncthreads();
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = v2f{ex_re[k1 * kExPitch + t], ex_im[k1 * kExPitch + t]};
      idft16v(v);
      __syncthreads();                    // the element buffer aliases the exchange planes
#pragma unroll
      for (int j = 0; j < 16; ++j)
        bufv[kElemPitch * (t + 16 * j) + a] = cmulv(cmulv(v[dft16_pos(j)], twa[j >> 2]), twb[j & 3]);
      __syncthreads();
#pragma unroll
      for (int aa = 0; aa < 16; ++aa) v[aa] = bufv[kElemPitch * tid + aa];
      idft16v(v);
      v2f w = rho ? w0[1] : w0[0];
      const v2f ws = rho ? wst[1] : wst[0];
#pragma unroll
      for (int ka = 0; ka < 16; ++ka) {
        v2f val = cmulv(v[dft16_pos(ka)], w);
        w = cmulv(w, ws);
        if (rho == 0) keep[ka] = val;
        else { const v2f k0 = keep[ka]; keep[ka] = k0 + val; v[dft16_pos(ka)] = k0 - val; }   // z_r + z_(r+2), z_r - z_(r+2)
      }
    }
    // first half hands a- over, second half b+
    __syncthreads();                      // every thread is done reading its element buffer
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) swap_mine[256 * ka + tid] = half == 0 ? v[dft16_pos(ka)] : keep[ka];
    __syncthreads();
    const FullbandSink sink = fullband_sink(out, slot, set.scale[sel], n_scales, row_len, seg, MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1);
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      const v2f o = swap_theirs[256 * ka + tid];
      const int64_t n2 = tid + 256 * ka;
      if (half == 0) {                    // a+ and b+
        const v2f ap = keep[ka];
        fullband_put<MODE>(sink, n2, make_float2(ap.x + o.x, ap.y + o.y));
        fullband_put<MODE>(sink, n2 + 2 * kRowLenDev, make_float2(ap.x - o.x, ap.y - o.y));
      } else {                            // a- and i b-
        const v2f bm = v[dft16_pos(ka)];
        fullband_put<MODE>(sink, n2 + kRowLenDev, make_float2(o.x - bm.y, o.y + bm.x));
        fullband_put<MODE>(sink, n2 + 3 * kRowLenDev, make_float2(o.x + bm.y, o.y - bm.x));
      }
    }
  }
}

// A tile's 16 columns are 64 bytes of an amplitude row: the tile next to it completes the 128-byte line,
// so neighbouring tiles go to workgroups 8 apart -- the same XCD, i.e. the same L2, a dispatch round apart
__device__ __forceinline__ int fullband_tile(int bid) { return (bid & ~15) | ((bid & 7) << 1) | ((bid >> 3) & 1); }

// len = 256: k_fft_cols256<+1, false> with the store above.  grid (ld / 16, slots), ld / 16 a multiple of 16
template <int MODE>
__global__ void __launch_bounds__(256, 4) k_fullband_cols256(const cf* __restrict__ in, float* __restrict__ out,
                                                          int ld, int64_t in_cstride,
                                                          const cf* __restrict__ tw256, int scale,
                                                          int n_scales, int64_t row_len, const SegOut seg) {
  __shared__ float ex_re[16 * kExColD];
  __shared__ float ex_im[16 * kExColD];
  const int c = blockIdx.y, col0 = fullband_tile(blockIdx.x) * 16, tid = threadIdx.x;
  const int s = tid & 15, t = tid >> 4;
  cf tw[16], v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) tw[j] = tw256[(t * j) & 255];
  const cf* x = in + (int64_t)c * in_cstride + col0 + s;
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = x[(int64_t)(t + 16 * j) * ld];
  fft256_16t<1>(v, tw, ex_re + s * kExColD, ex_im + s * kExColD, t);
  const FullbandSink sink = fullband_sink(out, c, scale, n_scales, row_len, seg, MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1);
#pragma unroll
  for (int j = 0; j < 16; ++j) fullband_put<MODE>(sink, (int64_t)(t + 16 * j) * ld + col0 + s, v[j]);
}

// len = 512, 1024: k_fft_colsq<+1, false, LQ> with the store above
template <int MODE, int LQ>
__global__ void __launch_bounds__(256, 2) k_fullband_colsq(const cf* __restrict__ in, float* __restrict__ out,
                                                        int ld, int64_t in_cstride,
                                                        const cf* __restrict__ tw4096,
                                                        const cf* __restrict__ tw256, int scale, int n_scales,
                                                        int64_t row_len, const SegOut seg) {
  constexpr int q = 1 << LQ, len = 256 * q;
  __shared__ float ex_re[16 * kExColD];
  __shared__ float ex_im[16 * kExColD];
  const int c = blockIdx.y, col0 = fullband_tile(blockIdx.x) * 16, tid = threadIdx.x;
  const int s = tid & 15, t = tid >> 4;
  __shared__ cf twl[256];
  twl[tid] = tw256[((tid & 15) * (tid >> 4)) & 255];
  constexpr int kInLds = q == 4 ? 1 : 0, kInReg = q - 1 - kInLds;
  __shared__ cf park[kInLds ? 16 * 256 : 1];
  cf v[16], u[kInReg * 16];
  const cf* x = in + (int64_t)c * in_cstride + col0 + s;
#pragma unroll
  for (int a = 0; a < q; ++a) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = x[(int64_t)(q * (t + 16 * j) + a) * ld];
    __syncthreads();
    fft256_16t_ldstw<1>(v, twl + t, ex_re + s * kExColD, ex_im + s * kExColD, t);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int kb = t + 16 * j;
      const cf val = cmul(v[j], tw4096_at<1>(tw4096, a * kb * (kRowLenDev / len)));
      if (a < kInLds) park[j * 256 + tid] = val;
      else if (a < q - 1) u[(a < q - 1 ? a - kInLds : 0) * 16 + j] = val;
      else v[j] = val;
    }
  }
  const FullbandSink sink = fullband_sink(out, c, scale, n_scales, row_len, seg, MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cf w[q];
#pragma unroll
    for (int a = 0; a < q - 1; ++a) w[a] = a < kInLds ? park[j * 256 + tid] : u[(a < kInLds ? 0 : a - kInLds) * 16 + j];
    w[q - 1] = v[j];
    dft_small<1, q>(w);
#pragma unroll
    for (int ka = 0; ka < q; ++ka)
      fullband_put<MODE>(sink, (int64_t)(t + 16 * j + 256 * ka) * ld + col0 + s, w[ka]);
  }
}

// ---- block convolution, the scales (kernels.h: BcBlocks; fwd64.hip: k_bc_forward makes the block spectra) ----
// One workgroup per (block, channel): the block's spectrum stays in registers while the group's scales go by --
// product with the scale's response (32 KB, L2: every workgroup reads the same ones), 4096-point inverse FFT
// (the arithmetic of k_fullband_rows), and the block's `hop` samples of the scale's row straight from the
// registers of the last DFT16, 256 consecutive samples per store.  HBM sees the spectrum once and the result.
// 152 registers under a launch bound of two waves per SIMD: three workgroups are resident (25 ms for 128 ch x 1e6
// x 83 scales; bound to three waves the compiler squeezed an earlier version into 150 registers and it took 48
// ms, held to two workgroups by LDS 33: profiles/r04_heavy_tails.md).  grid (blocks * channels)
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_bc_scales(const cf* __restrict__ xb, float* __restrict__ out,
                                                   const cf* __restrict__ h, const int32_t* __restrict__ rows,
                                                   int n_group_scales, const cf* __restrict__ twt,
                                                   const cf* __restrict__ tw256, const BcBlocks bl, int blk0,
                                                   int n_scales, int64_t col0, int64_t row_len,
                                                   const unsigned char* __restrict__ mask) {
  // twt[256 j + tid] = W_4096^(+(t + 16 j) a): the middle twiddles in the order the threads meet them
  constexpr int kElem = MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) cf buf[16 * kExColD];
  float* const ex_re = reinterpret_cast<float*>(buf) + (threadIdx.x & 15) * kExColD;
  float* const ex_im = ex_re + 16 * kExColD;
  __shared__ v2f twl[256];
  const int tid = threadIdx.x, a = tid & 15, t = tid >> 4;
  const int ch = blockIdx.x % bl.n_channels, lb = blockIdx.x / bl.n_channels, blk = blk0 + lb;
  int e = 0;
  while (e + 1 < bl.n_epochs && blk >= bl.blk_first[e + 1]) ++e;
  const int64_t n0 = (((bl.g_lo[e] / bl.hop) & ~(int64_t)1) + (blk - bl.blk_first[e])) * bl.hop;
  const int64_t lo = max(n0, bl.g_lo[e]), hi = min(n0 + bl.hop, bl.g_hi[e]);
  if (lo >= hi) return;                   // a block that only completes a pair of the forward transform
  // sample n of the recording is element n - (n0 - back) of the block: this thread's are tid + 256 ka
  const int first = (int)(lo - n0) + bl.back - tid, last = (int)(hi - n0) + bl.back - tid;   // first <= 256 ka < last
  {
    const cf w = tw256[(a * t) & 255];
    twl[tid] = v2f{w.x, w.y};
  }
  const int at = 16 * t + a;
  v2f xv[16], v[16];
  {
    const v2f* xp = reinterpret_cast<const v2f*>(xb) + ((int64_t)lb * bl.n_channels + ch) * kRowLenDev + at;
#pragma unroll
    for (int j = 0; j < 16; ++j) xv[j] = xp[256 * j];
  }
  const v2f* const tws = reinterpret_cast<const v2f*>(twt) + tid;
  v2f* const bufv = reinterpret_cast<v2f*>(buf);
  // middle twiddle j = twa[j >> 2] * twb[j & 3]: W^((t + 64 m) a) and W^(16 n a), eight values a thread keeps.
  // (All sixteen from the table inside the loop were sixteen L2 round trips per scale, each waited for in turn --
  // and a wait for a load is a wait for every store before it, one counter: 27.6 -> 25.6 ms for 83 scales.)
  v2f twa[4], twb[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) twa[m] = tws[256 * (4 * m)];
#pragma unroll
  for (int n = 0; n < 4; ++n) twb[n] = reinterpret_cast<const v2f*>(twt)[256 * n + a];   // t = 0: W^(16 n a)
  float* const o0 = out + ((int64_t)ch * n_scales * row_len + (n0 - bl.back + tid - col0)) * kElem;
  // The three radix-16 layers are synth_math.h's packed idft16v (a multiply by +-i is a register swizzle there:
  // written with float2 operators the loop spent a quarter of its instructions on moves); it leaves output k in
  // register dft16_pos(k).
  for (int s = 0; s < n_group_scales; ++s) {
    if (mask && !mask[rows[s]]) continue;   // masked run: the marked rows only (workgroup-uniform)
    const v2f* __restrict__ hs = reinterpret_cast<const v2f*>(h) + (int64_t)s * kRowLenDev + at;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmulv(xv[j], hs[256 * j]);
    __syncthreads();                      // twiddle table written / the buffer's last readers done
    idft16v(v);
#pragma unroll
    for (int m2 = 0; m2 < 16; ++m2) {
      const v2f u = cmulv(v[dft16_pos(m2)], twl[t + 16 * m2]);
      ex_re[t * kExPitch + m2] = u.x;
      ex_im[t * kExPitch + m2] = u.y;
    }
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = v2f{ex_re[k1 * kExPitch + t], ex_im[k1 * kExPitch + t]};
    idft16v(v);
    __syncthreads();                      // the element buffer aliases the exchange planes
#pragma unroll
    for (int j = 0; j < 16; ++j)
      bufv[kElemPitch * (t + 16 * j) + a] = cmulv(cmulv(v[dft16_pos(j)], twa[j >> 2]), twb[j & 3]);
    __syncthreads();
#pragma unroll
    for (int aa = 0; aa < 16; ++aa) v[aa] = bufv[kElemPitch * tid + aa];
    idft16v(v);
    float* const o = o0 + (int64_t)rows[s] * row_len * kElem;
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      if (256 * ka < first || 256 * ka >= last) continue;
      const v2f w = v[dft16_pos(ka)];
      if (MODE == GCWT_OUT_AMPLITUDE_F32) o[256 * ka] = __builtin_amdgcn_sqrtf(w.x * w.x + w.y * w.y);
      else if (MODE == GCWT_OUT_POWER_F32) o[256 * ka] = w.x * w.x + w.y * w.y;
      else reinterpret_cast<v2f*>(o)[256 * ka] = w;
    }
  }
}

// Forward column pass for REAL input and len = 256 q (q = 2, 4): the q interleaved subsequences of
// a column are real, so two of them share one FFT256 -- z = x_{2p} + i x_{2p+1}, split again with
// Z[kb] +- conj(Z[256 - kb]) -- before the W_len^(a kb) twiddles and the DFT_q: half the transforms
// of k_fft_colsq<-1, true, LQ>.  A thread keeps its own Z[kb] in registers and reads the partner
// rows from a [kb][column] tile in LDS (the last pair's tile aliases the exchange planes).
// grid (ld/16, slots); 39 KB (q = 2) or 74 KB (q = 4) of LDS
template <int LQ>
__global__ void __launch_bounds__(256, 2) k_fft_colsq_real2(const float* __restrict__ in, cf* __restrict__ out,
                                                         int ld, int64_t in_cstride, int64_t out_cstride,
                                                         int64_t tw_n, const cf* __restrict__ tw4096,
                                                         const cf* __restrict__ tw256,
                                                         const double* __restrict__ sums, double inv_n,
                                                         const SegIn segs, int rows_out) {
  constexpr int q = 1 << LQ, len = 256 * q, np = q / 2;
  static_assert(q == 2 || q == 4, "two or four subsequences");
  __shared__ __attribute__((aligned(16))) cf planes[16 * kExColD];       // exchange planes / last pair's tile
  __shared__ cf tile0[np > 1 ? 256 * 17 : 1];                           // first pair's tile (q = 4)
  __shared__ cf twl[256];                 // W256^(-t j) at [j][t]
  float* const ex_re = reinterpret_cast<float*>(planes);
  float* const ex_im = ex_re + 16 * kExColD;
  const int c = blockIdx.y, col0 = blockIdx.x * 16, tid = threadIdx.x;   // c: workspace slot
  const int s = tid & 15, t = tid >> 4;
  {
    const cf w = tw256[((tid & 15) * (tid >> 4)) & 255];
    twl[tid] = make_float2(w.x, -w.y);
  }
  const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
  const int64_t n_valid = segs.n_valid[g], n_lead = segs.n_lead[g];
  const float* x = in + (int64_t)ch * in_cstride + segs.x_off[g];
  const double mean = sums[ch] * inv_n;     // subtracted in fp64
  cf v[16], u[np > 1 ? 16 : 1];
#pragma unroll
  for (int p = 0; p < np; ++p) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t na = (int64_t)(q * (t + 16 * j) + 2 * p) * ld + col0 + s, nb = na + ld;
      const float xa = x[min(max(na, n_lead), n_valid - 1)], xb = x[min(max(nb, n_lead), n_valid - 1)];   // clamped
      v[j] = make_float2(na >= n_lead && na < n_valid ? (float)((double)xa - mean) : 0.f,
                         nb >= n_lead && nb < n_valid ? (float)((double)xb - mean) : 0.f);
    }
    __syncthreads();                      // twiddle table written / previous exchange read
    fft256_16t_ldstw<-1>(v, twl + t, ex_re + s * kExColD, ex_im + s * kExColD, t);
    if (p < np - 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { tile0[(t + 16 * j) * 17 + s] = v[j]; u[j] = v[j]; }
    }
  }
  __syncthreads();                        // the last pair's tile aliases the planes
#pragma unroll
  for (int j = 0; j < 16; ++j) planes[(t + 16 * j) * 17 + s] = v[j];
  __syncthreads();
  cf* o = out + (int64_t)c * out_cstride + col0 + s;
  cf stq = make_float2(1.f, 0.f), st16 = stq, wj = stq;
  if (tw_n > 0) {
    stq = unit_phase((int64_t)(col0 + s) * 256, tw_n, -1);
    st16 = unit_phase((int64_t)(col0 + s) * 16, tw_n, -1);
    wj = unit_phase((int64_t)(col0 + s) * t, tw_n, -1);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kb = t + 16 * j, km = (256 - kb) & 255;
    cf w[q];
#pragma unroll
    for (int p = 0; p < np; ++p) {
      const cf zk = p < np - 1 ? u[j] : v[j];
      const cf zm = p < np - 1 ? tile0[km * 17 + s] : planes[km * 17 + s];
      const cf xa = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const cf xb = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
      w[2 * p] = p == 0 ? xa : cmul(xa, tw4096_at<-1>(tw4096, 2 * p * kb * (kRowLenDev / len)));
      w[2 * p + 1] = cmul(xb, tw4096_at<-1>(tw4096, (2 * p + 1) * kb * (kRowLenDev / len)));
    }
    dft_small<-1, q>(w);
    cf ph = wj;
#pragma unroll
    for (int ka = 0; ka < q; ++ka) {
      const int k = kb + 256 * ka;
      if (k < rows_out) {
        cf val = w[ka];
        if (tw_n > 0) val = cmul(val, ph);
        o[(int64_t)k * ld] = val;
      }
      if (tw_n > 0) ph = cmul(ph, stq);
    }
    if (tw_n > 0) wj = cmul(wj, st16);
  }
}

// Forward column pass for REAL input, two columns per FFT: z = x[.., 2m] + i x[.., 2m+1]
// goes through one FFT256 and is split again with Z[k] +- conj(Z[256-k]); only rows
// k = 0 .. 128 exist afterwards (the mirrored rows are reflected by the row pass).
// Tile of 32 real columns per workgroup: 128-byte reads, 256-byte writes.  grid (ld/32, C)
__global__ void __launch_bounds__(256) k_fft_cols256_real2(const float* __restrict__ in, cf* __restrict__ out,
                                                           int ld, int64_t in_cstride, int64_t out_cstride,
                                                           int64_t tw_n, const cf* __restrict__ tw256,
                                                           const double* __restrict__ sums, double inv_n,
                                                           const SegIn segs) {
  // exchange planes (lanes over columns: 290-float column stride) aliased by the 256 x 17
  // tile the split stage reads
  __shared__ __attribute__((aligned(16))) cf tile[16 * kExColD];
  float* const ex_re = reinterpret_cast<float*>(tile);
  float* const ex_im = ex_re + 16 * kExColD;
  const int c = blockIdx.y, col0 = blockIdx.x * 32, tid = threadIdx.x;   // c: workspace slot
  const int g = c / segs.n_channels, ch = c - g * segs.n_channels;
  const int64_t n_valid = segs.n_valid[g], n_lead = segs.n_lead[g];
  const float* x = in + (int64_t)ch * in_cstride + segs.x_off[g];
  const double mean = sums[ch] * inv_n;     // subtracted in fp64 (transforms.py:142-143 works in float64)
  const int s = tid & 15, t = tid >> 4;
  cf tw[16], v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const cf w = tw256[(t * j) & 255];
    tw[j] = make_float2(w.x, -w.y);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t n = (int64_t)(t + 16 * j) * ld + col0 + 2 * s;
    const int64_t na = min(max(n, n_lead), n_valid - 1), nb = min(max(n + 1, n_lead), n_valid - 1);
    const float a = x[na], b = x[nb];               // clamped: no branch around the loads
    v[j] = make_float2(n >= n_lead && n < n_valid ? (float)((double)a - mean) : 0.f,
                       n + 1 >= n_lead && n + 1 < n_valid ? (float)((double)b - mean) : 0.f);
  }
  fft256_16t<-1>(v, tw, ex_re + s * kExColD, ex_im + s * kExColD, t);
  __syncthreads();   // the tile aliases the planes
#pragma unroll
  for (int j = 0; j < 16; ++j) tile[(t + 16 * j) * 17 + s] = v[j];
  __syncthreads();
  // split and twiddle: thread = one real column, rows k = k0 + 8 i
  const int cr = tid & 31, k0 = tid >> 5, m = cr >> 1;
  const bool odd = cr & 1;
  cf w = make_float2(1.f, 0.f), st = w;
  if (tw_n > 0) {
    w = unit_phase((int64_t)(col0 + cr) * k0, tw_n, -1);
    st = unit_phase((int64_t)(col0 + cr) * 8, tw_n, -1);
  }
  cf* o = out + (int64_t)c * out_cstride + col0 + cr;
  for (int k = k0; k <= 128; k += 8) {
    const cf zk = tile[k * 17 + m], zm = tile[((256 - k) & 255) * 17 + m];
    const cf val = odd ? make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x))
                       : make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    o[(int64_t)k * ld] = cmul(val, w);
    w = cmul(w, st);
  }
}

// ---------------------------------------------------------------------------
// Block spectra: XB[blk][k] = scale * FFT_256(x_R[(blk*hop - halo + n) mod M])
// 16 blocks per workgroup, 16 threads per block.  grid (ceil(nblk/16), C)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_block_fft(const cf* __restrict__ xr, cf* __restrict__ xb,
                                                   int64_t m_mask, int hop, int halo, int blk_lo, int nblk,
                                                   int64_t xr_cstride, int64_t xb_cstride,
                                                   const cf* __restrict__ tw256, float scale) {
  __shared__ float ex_re[16 * kExCol];
  __shared__ float ex_im[16 * kExCol];
  const int c = blockIdx.y;
  const int colw = threadIdx.x >> 4, t = threadIdx.x & 15;
  const int blk = blockIdx.x * 16 + colw;
  const bool valid = blk < nblk;
  cf tw[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    cf w = tw256[(t * j) & 255];
    tw[j] = make_float2(w.x, -w.y);  // table holds exp(+2 pi i q/256)
  }
  cf v[16];
  const cf* x = xr + (int64_t)c * xr_cstride;
  const int64_t base = (int64_t)(blk_lo + blk) * hop - halo + t;   // block blk_lo + blk -> XB[blk]
#pragma unroll
  for (int j = 0; j < 16; ++j)
    v[j] = valid ? x[(base + 16 * j) & m_mask] : make_float2(0.f, 0.f);
  fft256_16t<-1>(v, tw, ex_re + colw * kExCol, ex_im + colw * kExCol, t);
  if (valid) {
    cf* o = xb + (int64_t)c * xb_cstride + (int64_t)blk * 256 + t;
#pragma unroll
    for (int j = 0; j < 16; ++j) o[16 * j] = make_float2(v[j].x * scale, v[j].y * scale);
  }
}

// ---------------------------------------------------------------------------
// Synthesis (the hot kernel).  One workgroup = one (channel, scale, group of
// blocks).  Columns are (block, r) pairs, r in [0, R): column (blk, r) is
//   y[R (blk*hop + m - halo) + r] = IFFT_256( XB_blk[k] H_s[k] W_{256 R}^{k r} )[m]
// kept for halo <= m < halo + hop.  16 columns per batch, 16 threads per column;
// amplitudes are staged in LDS and stored as contiguous runs.
// grid (n_items, C)
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) k_synth(const SynthArgs a) {
  __shared__ float ex_re[16 * kExCol];
  __shared__ float ex_im[16 * kExCol];
  __shared__ __attribute__((aligned(16))) float tile[(MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1) * 16 * 256];
  constexpr int kElem = MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1;  // floats per output sample

  const SynthItemDev it = a.items[blockIdx.x];
  const SynthLevelDev lv = a.levels[it.level];
  const int c = blockIdx.y;               // workspace slot: segment * n_channels + channel
  const int seg = c / a.seg.n_channels, ch = c - seg * a.seg.n_channels;
  const int64_t w_lo = a.seg.w_lo[seg], w_hi = a.seg.w_hi[seg];
  const int R = lv.decimation, hop = lv.hop, halo = lv.halo;
  {
    // union grids of a batch: nothing to do when these blocks keep no sample in the window
    const int64_t span = (int64_t)hop * R;
    const int64_t first = (int64_t)(lv.blk_base + it.blk0) * span;
    if (first + (int64_t)it.nblk * span <= w_lo || first >= w_hi) return;
  }
  const int colw = threadIdx.x >> 4, t = threadIdx.x & 15;

  cf tw[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) tw[j] = a.tw256[(t * j) & 255];

  cf hk[16];
  const cf* bank = a.bank + (int64_t)it.scale * 256 + t;
#pragma unroll
  for (int j = 0; j < 16; ++j) hk[j] = bank[16 * j];

  const cf* xb = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset;
  const cf* ltw = a.level_tw + lv.tw_offset;
  float* out = a.out + (((int64_t)ch * a.n_scales + it.scale) * a.row_len + a.seg.seg_col[seg]) * kElem;

  const int ncols = it.nblk * R;
  // per batch geometry of the staged tile
  const int row_stride = R <= 16 ? 16 : R;  // global distance (samples) between tile rows of 16
  for (int col0 = 0; col0 < ncols; col0 += 16) {
    const int col = col0 + colw;
    const bool valid = col < ncols;
    const int blk_l = col / R;           // block within the item
    const int r = col - blk_l * R;
    const int blk = it.blk0 + blk_l;     // index into the level's computed blocks

    cf v[16];
    if (valid) {
      const cf* xbp = xb + (int64_t)blk * 256 + t;
      // W^{(t + 16 j) r} = W^{t r} (W^{16 r})^j, W = exp(2 pi i/(256 R)); both factors
      // come from the level table (indices t r and 16 r are below 16 R)
      cf wcur = ltw[t * r];
      const cf wstep = ltw[16 * r];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] = cmul(cmul(xbp[16 * j], hk[j]), wcur);
        wcur = cmul(wcur, wstep);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = make_float2(0.f, 0.f);
    }
    fft256_16t<1>(v, tw, ex_re + colw * kExCol, ex_im + colw * kExCol, t);

    // stage: sample (m = t + 16 j, r) -> tile
    const int blk_b = (col0 / R);        // first block (within item) of this batch when R <= 16
    if (valid) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int m = t + 16 * j - halo;
        if (m >= 0 && m < hop) {
          int q;
          if (R <= 16) q = ((blk_l - blk_b) * hop + m) * R + r;
          else q = m * 16 + (r & 15);
          if (MODE == GCWT_OUT_AMPLITUDE_F32) tile[q] = sqrtf(v[j].x * v[j].x + v[j].y * v[j].y);
          else if (MODE == GCWT_OUT_POWER_F32) tile[q] = v[j].x * v[j].x + v[j].y * v[j].y;
          else { tile[2 * q] = v[j].x; tile[2 * q + 1] = v[j].y; }
        }
      }
    }
    __syncthreads();
    // copy out: tile element q lives at global sample n(q) = n0 + (q/16)*row_stride + q%16
    {
      int64_t n0;
      int count;  // staged samples
      if (R <= 16) {
        const int nb = min(16 / R, it.nblk - blk_b);
        n0 = (int64_t)(lv.blk_base + it.blk0 + blk_b) * hop * R;
        count = nb * hop * R;
      } else {
        const int blk_cur = lv.blk_base + it.blk0 + col0 / R;
        const int r0 = col0 % R;
        n0 = (int64_t)blk_cur * hop * R + r0;
        count = hop * 16;
      }
      for (int q = threadIdx.x; q < count; q += 256) {
        const int64_t n = n0 + (int64_t)(q >> 4) * row_stride + (q & 15);
        if (n >= w_lo && n < w_hi) {
          if (kElem == 1) out[n] = tile[q];
          else { out[2 * n] = tile[2 * q]; out[2 * n + 1] = tile[2 * q + 1]; }
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Direct scales: W[n] = sum_j (x[n + (L-1)/2 - j] - mean) psi[j] inside the epoch
// (convolution.py:68-87 'same' crop; transforms.py:202-204), for kernels of up to
// kDirectMaxLen taps -- evaluated by parts (k_build_direct): the tile holds the first DIFFERENCE
// d[m] = x~[m] - x~[m-1] of the mean-removed, zero-extended epoch (x[m] - x[m-1] inside it: exact or
// rounded relative to the increment; x~[0] and -x~[Ne-1] at its two ends), the taps are the running
// sums of psi, and the kernel's zero-frequency response times x~[q - L + 1] is added at the end.
//
// One workgroup owns kDirectTile consecutive output samples of one (epoch, channel): it
// parks the samples those outputs need (tile + half the longest kernel on either side,
// mean removed, zero outside the epoch) in LDS once and walks over ALL direct scales.
// A thread computes 8 consecutive outputs as 8 (re, im) accumulator pairs; taps go in groups
// of 8, and a group needs the 15 samples x[b .. b+14], b = first output + top - 8 g - 7.
// The kernel of a scale is stored behind `front` = (7 - top) mod 8 zero taps
// (direct_front_pad), which makes b a multiple of 8 for every thread and group: the window is
// two aligned 8-sample chunks, the lower one new in each group (two ds_read_b128 with no
// address arithmetic; chunks sit 12 words apart, so the 16 lanes of a b128 access hit
// distinct banks) and the upper one the previous group's lower chunk (two groups per trip,
// roles swapped, no copies).  A tap is wave-uniform -- an SGPR pair from a scalar load -- and
// one v_pk_fma_f32 adds sample * (re, im) to an accumulator pair, the sample broadcast to
// both halves by op_sel: 64 packed FMAs per 8 taps and nothing else on the vector pipe.
// Results leave through a wave-private LDS transposition, so that a wave stores 1 KB runs
// (dwordx4 per lane, or 256 B runs of dwords when the row is not 16-byte aligned there)
// instead of 32-byte pieces per lane.
// psi of a scale: front zeros + L taps + zeros up to a multiple of 8 (+8).
// grid (ceil(longest range / kDirectTile), 1, n_epochs * C): up to 16 epochs per launch
// ---------------------------------------------------------------------------
constexpr int kDirectTile = 2048;     // outputs per workgroup: 256 threads x 8

__device__ __forceinline__ int direct_chunk(int c) { return 12 * c; }   // word offset of chunk c
__device__ __forceinline__ int direct_stage(int f) { return f + ((f >> 6) << 2); }

template <int HI>   // acc += w[HI] * k, w = one of the two samples of the register pair
__device__ __forceinline__ void direct_fma(v2f& acc, v2f w, v2f k) {
  if constexpr (HI == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "s"(k));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "s"(k));
}

// 8 taps x 8 outputs: window element i (0..14) is lo[i] for i < 8, hi[i - 8] above
__device__ __forceinline__ void direct_group(v2f (&acc)[8], const v4f& l0, const v4f& l1, const v4f& h0,
                                             const v4f& h1, const cf* __restrict__ taps) {
  const v2f wp[8] = {{l0.x, l0.y}, {l0.z, l0.w}, {l1.x, l1.y}, {l1.z, l1.w},
                     {h0.x, h0.y}, {h0.z, h0.w}, {h1.x, h1.y}, {h1.z, h1.w}};
  v2f k[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { const cf t = taps[u]; k[u] = v2f{t.x, t.y}; }   // wave-uniform: scalar loads
#pragma unroll
  for (int u = 0; u < 8; ++u) {
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int i = 7 - u + o;
      if ((i & 1) == 0) direct_fma<0>(acc[o], wp[i >> 1], k[u]);
      else direct_fma<1>(acc[o], wp[i >> 1], k[u]);
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_direct(const float* __restrict__ x, float* __restrict__ out,
                                                const cf* __restrict__ psi,
                                                const DirectScale* __restrict__ sc, int n_direct,
                                                const double* __restrict__ sums, double inv_n,
                                                int64_t n_samples, int n_scales,
                                                const DirectEpochs eps, int64_t col0,
                                                int64_t row_len, int halo, const cf* __restrict__ tail,
                                                const unsigned char* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kElem = MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1;
  constexpr int kStageWave = 512 * kElem + 32 * kElem;        // floats of one wave's transposition area
  float* const tile = reinterpret_cast<float*>(smem);
  const int e = blockIdx.z / eps.n_channels, c = blockIdx.z - e * eps.n_channels;
  const int64_t epoch_start = eps.epoch_start[e], epoch_len = eps.epoch_len[e];
  const int64_t g0 = eps.g_lo[e] + (int64_t)blockIdx.x * kDirectTile;   // first output, recording index
  if (g0 >= eps.g_hi[e]) return;
  const int64_t base = g0 - epoch_start;                               // epoch-local index of output 0
  const double mean = sums[c] * inv_n;      // subtracted in fp64
  const float* xe = x + (int64_t)c * n_samples + epoch_start;
  // sample base - halo + i of the epoch is element i & 7 of chunk i >> 3
  const int n_tile = kDirectTile + 2 * halo;
  float* const xt = tile + direct_chunk(n_tile >> 3) + 4 * kStageWave;   // x~ itself, plain layout (the tail term)
  for (int i = threadIdx.x; i < n_tile; i += 256) {
    const int64_t m = base - halo + i;
    const bool in = m >= 0 && m < epoch_len;
    const float xv = in ? (float)((double)xe[m] - mean) : 0.f;
    float d = 0.f;
    if (m >= 1 && m < epoch_len) d = xe[m] - xe[m - 1];        // the mean cancels
    else if (m == 0) d = xv;                                      // x~[0] - 0
    else if (m == epoch_len) d = -(float)((double)xe[m - 1] - mean);   // 0 - x~[Ne - 1]
    tile[direct_chunk(i >> 3) + (i & 7)] = d;
    xt[i] = xv;
  }
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const stage = tile + direct_chunk(n_tile >> 3) + wave * kStageWave;
  const int64_t wave_left = (eps.g_hi[e] - g0 - 512 * wave) * kElem;    // floats of this wave inside the range
  for (int d = 0; d < n_direct; ++d) {
    const DirectScale p = sc[d];
    if (mask && !mask[p.scale]) continue;        // a masked run (api.cpp: reroute_scales) makes the marked rows only
    const cf* taps = psi + p.offset;
    const int top = (int)((p.length - 1) / 2) + p.front;            // = 7 (mod 8)
    const int n_groups = (int)((p.length + p.front + 7) >> 3);
    // output o of this thread reads sample q0 + o - j of the tile for tap j; q0 + 1 starts a chunk
    const float* w_ptr = tile + direct_chunk((8 * tid + top + halo + 1) >> 3);
    v2f acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = v2f{0.f, 0.f};
    v4f a0, a1;
    v4f b0 = *reinterpret_cast<const v4f*>(w_ptr), b1 = *reinterpret_cast<const v4f*>(w_ptr + 4);
    for (int g = 0; g < n_groups; g += 2) {
      w_ptr -= 12;
      a0 = *reinterpret_cast<const v4f*>(w_ptr);
      a1 = *reinterpret_cast<const v4f*>(w_ptr + 4);
      direct_group(acc, a0, a1, b0, b1, taps);
      if (g + 1 < n_groups) {
        w_ptr -= 12;
        b0 = *reinterpret_cast<const v4f*>(w_ptr);
        b1 = *reinterpret_cast<const v4f*>(w_ptr + 4);
        direct_group(acc, b0, b1, a0, a1, taps + 8);
      }
      taps += 16;
    }
    {   // + Psi[L-1] x~[q - L + 1]: the kernel's response at zero frequency times the signal itself
      const cf tl = tail[d];
      const float* xq = xt + halo + 8 * tid + (int)((p.length - 1) / 2) - (int)(p.length - 1);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const float xv = xq[o];
        acc[o] = v2f{fmaf(xv, tl.x, acc[o].x), fmaf(xv, tl.y, acc[o].y)};
      }
    }
    // this thread's 8 * kElem floats, then the wave's 512 * kElem floats in runs of 4 per lane
    float v[8 * kElem];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      if (MODE == GCWT_OUT_AMPLITUDE_F32) v[o] = sqrtf(acc[o].x * acc[o].x + acc[o].y * acc[o].y);
      else if (MODE == GCWT_OUT_POWER_F32) v[o] = acc[o].x * acc[o].x + acc[o].y * acc[o].y;
      else { v[2 * o] = acc[o].x; v[2 * o + 1] = acc[o].y; }
    }
#pragma unroll
    for (int i = 0; i < 2 * kElem; ++i)
      *reinterpret_cast<v4f*>(stage + direct_stage(lane * 8 * kElem + 4 * i)) =
          v4f{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* const o_row = out + (((int64_t)c * n_scales + p.scale) * row_len + (g0 - col0) + 512 * wave) * kElem;
    const bool aligned = (reinterpret_cast<uintptr_t>(o_row) & 15) == 0;     // wave-uniform
    if (aligned) {
#pragma unroll
      for (int i = 0; i < 2 * kElem; ++i) {
        const int f = 256 * i + 4 * lane;
        const v4f r = *reinterpret_cast<const v4f*>(stage + direct_stage(f));
        if (f + 4 <= wave_left) __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(o_row + f));
        else {
          if (f < wave_left) o_row[f] = r.x;
          if (f + 1 < wave_left) o_row[f + 1] = r.y;
          if (f + 2 < wave_left) o_row[f + 2] = r.z;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8 * kElem; ++i) {
        const int f = 64 * i + lane;
        const float r = stage[direct_stage(f)];
        if (f < wave_left) __builtin_nontemporal_store(r, o_row + f);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// Level IFFT for decimations above 256 (M = P/R <= 8192 points): one workgroup per
// channel gathers X~[j1][0:q], j1 < n1, applies the length-q row DFT and the
// W_M^(j1 m2) twiddle directly, and runs the length-n1 column FFTs in LDS.
//   x_R[q m1 + m2] = sum_j1 e^{2 pi i j1 m1/n1} e^{2 pi i j1 m2/M} sum_j2 X~[j1][j2] e^{2 pi i j2 m2/q}
// grid (C), dynamic LDS n1*q*8 bytes
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_level_small(const cf* __restrict__ x, cf* __restrict__ xr,
                                                     int n1, int log2n1, int q, int64_t row_stride,
                                                     int64_t x_cstride, int64_t xr_cstride,
                                                     const cf* __restrict__ tw4096, const RowTaper tp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* buf = reinterpret_cast<cf*>(smem);
  const cf* xc = x + (int64_t)blockIdx.x * x_cstride;
  const int m = n1 * q;
  for (int idx = threadIdx.x; idx < m; idx += 256) {
    const int j1 = idx / q, m2 = idx - j1 * q;
    cf acc = make_float2(0.f, 0.f);
    for (int j2 = 0; j2 < q; ++j2) {          // q <= 4096, a power of two: the phases come from the table
      cf v = xc[(int64_t)j1 * row_stride + j2];
      if (tp.p1) {
        const float f = row_taper(tp, j1, j2);
        v = make_float2(v.x * f, v.y * f);
      }
      acc = cadd(acc, cmul(v, tw4096_at<1>(tw4096, ((j2 * m2) & (q - 1)) * (kRowLenDev / q))));
    }
    buf[idx] = cmul(acc, unit_phase((int64_t)j1 * m2, m, 1));
  }
  __syncthreads();
  if (n1 > 1) lds_fft_radix2<1>(buf, n1, log2n1, q, 1, q, tw4096);
  cf* o = xr + (int64_t)blockIdx.x * xr_cstride;
  for (int idx = threadIdx.x; idx < m; idx += 256) {
    const int m1 = idx / q, m2 = idx - m1 * q;
    o[idx] = buf[bitrev(m1, log2n1) * q + m2];
  }
}

// ---------------------------------------------------------------------------
// Bluestein (chirp-z) DFT of arbitrary length N on top of the power-of-two FFTs:
//   DFT_N(v)[k] = W[k] * sum_n (v[n] W[n]) conj(W)[k-n],   W[n] = exp(-i pi n^2 / N)
// (the identity ghost/sigtools/fourier.py:9-52 uses).  n^2 is reduced mod 2N in
// integers so the chirp phase is exact before the fp64 sincos.
__device__ __forceinline__ cf chirp_w(int64_t n, int64_t N, double sign) {
  const int64_t t = (n * n) % (2 * N);
  double s, c;
  sincospi(sign * (double)t / (double)N, &s, &c);
  return make_float2((float)c, (float)s);
}

// Circular chirp kernel b[j] = conj(W)[d], d = j (j < N) or P - j (P - j < N), else 0.
__global__ void k_chirp_kernel(cf* __restrict__ b, int64_t N, int64_t P) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= P) return;
  const int64_t d = j < N ? j : (P - j < N ? P - j : -1);
  b[j] = d < 0 ? make_float2(0.f, 0.f) : chirp_w(d, N, +1.0);
}

// a[n] = (v[n] - m) W[n], n < N (v = 0 beyond n_valid), 0 for N <= n < P.  v is real or
// complex (optionally conjugated: the inverse DFT is conj(DFT(conj v))/N); m = sums[0]*inv_n.
__global__ void k_chirp_load(const float* __restrict__ v, int is_complex, int conj_in,
                             int64_t n_valid, int64_t N, int64_t P,
                             const double* __restrict__ sums, double inv_n, cf* __restrict__ a) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= P) return;
  cf z = make_float2(0.f, 0.f);
  if (n < N) {
    const float m = (float)(sums[0] * inv_n);
    if (n < n_valid) {
      if (is_complex) {
        z = reinterpret_cast<const cf*>(v)[n];
        if (conj_in) z.y = -z.y;
      } else {
        z.x = v[n];
      }
    }
    z.x -= m;
    z = cmul(z, chirp_w(n, N, -1.0));
  }
  a[n] = z;
}

// Second leg of the analytic signal: with y = conv(a, b) the spectrum is W[k] y[k]/P and the
// inverse DFT's chirped input conj(mask X) W collapses to mask[k] conj(y[k]) / P (|W| = 1).
// mask = 1 at DC (and at N/2 for even N), 2 on the other non-negative bins, 0 elsewhere
// (ghost/sigtools/analytic.py:100-108).  In place.
__global__ void k_chirp_analytic_mask(cf* __restrict__ y, int64_t N, int64_t P, float scale) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  float g = 0.f;
  if (k < N) {
    if (k == 0 || ((N & 1) == 0 && k == N / 2)) g = scale;
    else if (k < (N + 1) / 2) g = 2.f * scale;
  }
  const cf v = y[k];
  y[k] = make_float2(g * v.x, -g * v.y);
}

// out[n] = scale * W[n] y[n] (conjugated when conj_out) + add_re, n < count.
__global__ void k_chirp_store(const cf* __restrict__ y, cf* __restrict__ out, int64_t count,
                              int64_t N, float scale, int conj_out,
                              const double* __restrict__ sums, double inv_n) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= count) return;
  cf z = cmul(y[n], chirp_w(n, N, -1.0));
  z.x *= scale;
  z.y *= conj_out ? -scale : scale;
  z.x += (float)(sums[0] * inv_n);
  out[n] = z;
}

// a[i] *= b[i] (complex), n elements.  grid (ceil(n/256))
__global__ void k_cmul_inplace(cf* __restrict__ a, const cf* __restrict__ b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = cmul(a[i], b[i]);
}

// zero [start, stop) of every (channel, scale) row.  grid (ceil(len/256), C*S)
__global__ void k_zero_range(float* __restrict__ out, int64_t row_len_floats, int64_t start,
                             int64_t len) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < len) out[(int64_t)blockIdx.y * row_len_floats + start + i] = 0.f;
}

// ---------------------------------------------------------------------------
// Shifted band of a level (heavy-tailed wavelets: planner.h, LevelPlan::band_shift): the M = P1 q
// spectrum samples X[k - U], k in [0, M), U = P1 u2 of them below zero frequency, gathered from
// the k1-major positive-frequency spectrum X~[k1][k2] = X[k1 + P1 k2] of a REAL signal into
// Xs[k1][j2] = X[k1 + P1 (j2 - u2)]:  j2 >= u2 reads X~[k1][j2 - u2]; below, X[-n] = conj(X[n]) with
// n = P1 (u2 - j2) - k1 = (P1 - k1) + P1 (u2 - j2 - 1) for k1 > 0.  The level's row and column
// passes then run on Xs as they do on X~.
// grid (ceil(q / 256), P1, slots), block 256
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_shift_gather(const cf* __restrict__ x, cf* __restrict__ xs, int p1, int q,
                                                      int u2, int64_t x_row, int64_t x_cstride, int64_t xs_cstride) {
  const int j2 = blockIdx.x * 256 + threadIdx.x, k1 = blockIdx.y;
  if (j2 >= q) return;
  const cf* xc = x + (int64_t)blockIdx.z * x_cstride;
  cf v;
  if (j2 >= u2) {
    v = xc[(int64_t)k1 * x_row + (j2 - u2)];
  } else {
    const int row = k1 == 0 ? 0 : p1 - k1, col = u2 - j2 - (k1 == 0 ? 0 : 1);
    v = xc[(int64_t)row * x_row + col];
    v.y = -v.y;
  }
  xs[(int64_t)blockIdx.z * xs_cstride + (int64_t)k1 * q + j2] = v;
}

// ===========================================================================
// launch wrappers
// ===========================================================================
#define GCWT_LAUNCH_CHECK()                      \
  do {                                           \
    hipError_t e_ = hipGetLastError();           \
    if (e_ != hipSuccess) return e_;             \
  } while (0)

static int ilog2(int64_t v) { int l = 0; while ((1LL << l) < v) ++l; return l; }

hipError_t launch_channel_sum(const float* x, int64_t n, int n_channels, double* sums,
                              hipStream_t st) {
  int parts = (int)std::min<int64_t>(kSumPartsOwn, (n + 256 * 32 - 1) / (256 * 32));
  if (parts < 1) parts = 1;
  hipLaunchKernelGGL(k_channel_sum, dim3(parts, n_channels), dim3(256), 0, st, x, n, sums);
  GCWT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_channel_sum_final, dim3((n_channels + 63) / 64), dim3(64), 0, st, sums, n_channels, parts);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_channel_sum_final(double* sums, int n_channels, int parts, hipStream_t st) {
  if (parts < 1 || parts > kSumParts) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_channel_sum_final, dim3((n_channels + 63) / 64), dim3(64), 0, st, sums, n_channels, parts);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_shift_gather(const cf* x, cf* xs, int p1, int q, int u2, int64_t x_row, int64_t x_cstride,
                               int64_t xs_cstride, int n_channels, hipStream_t st) {
  hipLaunchKernelGGL(k_shift_gather, dim3((q + 255) / 256, p1, n_channels), dim3(256), 0, st, x, xs, p1, q, u2,
                     x_row, x_cstride, xs_cstride);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_build_bank(cf* bank, float* gain, const BankScale* sc, const double* amps,
                             int n_scales, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_build_bank, dim3(n_scales), dim3(B), 0, st, bank, gain, sc, amps, B);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_bank_gain(const cf* bank, float* gain, const BankScale* sc, int n_scales,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_bank_gain, dim3(n_scales), dim3(256), 0, st, bank, gain, sc);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_scale_windows(const float* gain, int32_t* scale_list, int n_listed, float tol,
                                float* gain_lv, bool prune, hipStream_t st) {
  if (n_listed <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scale_windows, dim3(n_listed), dim3(256), 0, st, gain, scale_list, tol, gain_lv,
                     prune ? 1 : 0);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_build_direct(cf* psi, const DirectScale* sc, int n_direct, int64_t max_len,
                               const double* amps, cf* tail, cf* psi_literal, hipStream_t st) {
  if (n_direct == 0) return hipSuccess;
  (void)max_len;
  hipLaunchKernelGGL(k_build_direct, dim3(n_direct), dim3(256), 0, st, psi, sc, amps, tail, psi_literal);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

static hipError_t launch_fft_cols_segs(int sign, bool real_in, const void* in, cf* out, int len, int ld,
                                       int64_t in_cstride, int64_t out_cstride, int64_t tw_n,
                                       const cf* tw4096, const cf* tw256, const double* sums,
                                       double inv_n, const SegIn& segs, int n_segments, hipStream_t st,
                                       int rows_out);

template <int SIGN, bool REAL_IN, int LQ>
static hipError_t launch_colsq(const void* in, cf* out, int ld, int64_t in_cstride, int64_t out_cstride,
                               int64_t tw_n, const cf* tw4096, const cf* tw256, const double* sums,
                               double inv_n, const SegIn& segs, int n_slots, int rows_out, hipStream_t st) {
  hipLaunchKernelGGL((k_fft_colsq<SIGN, REAL_IN, LQ>), dim3(ld / 16, n_slots), dim3(256), 0, st, in, out,
                     ld, in_cstride, out_cstride, tw_n, tw4096, tw256, sums, inv_n, segs, rows_out);
  return hipGetLastError();
}

hipError_t launch_fft_cols(int sign, bool real_in, const void* in, cf* out, int len, int ld,
                           int64_t in_cstride, int64_t out_cstride, int64_t tw_n, const cf* tw4096,
                           const cf* tw256,
                           const double* sums, double inv_n, int64_t n_valid, int n_channels,
                           hipStream_t st, int64_t n_lead, int rows_out) {
  SegIn one{};
  one.n_valid[0] = n_valid;
  one.n_lead[0] = n_lead;
  one.n_channels = n_channels;
  return launch_fft_cols_segs(sign, real_in, in, out, len, ld, in_cstride, out_cstride, tw_n, tw4096,
                              tw256, sums, inv_n, one, 1, st, rows_out);
}

hipError_t launch_fft_cols_batch(const float* in, cf* out, int len, int ld, int64_t in_cstride,
                                 int64_t out_cstride, int64_t tw_n, const cf* tw4096, const cf* tw256,
                                 const double* sums, double inv_n, const SegIn& segs, int n_segments,
                                 hipStream_t st, int rows_out) {
  return launch_fft_cols_segs(-1, true, in, out, len, ld, in_cstride, out_cstride, tw_n, tw4096, tw256,
                              sums, inv_n, segs, n_segments, st, rows_out);
}

static hipError_t launch_fft_cols_segs(int sign, bool real_in, const void* in, cf* out, int len, int ld,
                                       int64_t in_cstride, int64_t out_cstride, int64_t tw_n,
                                       const cf* tw4096, const cf* tw256, const double* sums,
                                       double inv_n, const SegIn& segs, int n_segments, hipStream_t st,
                                       int rows_out) {
  const int n_channels = segs.n_channels * n_segments;   // workspace slots
  bool all_valid = true;
  for (int g = 0; g < n_segments; ++g) all_valid = all_valid && segs.n_valid[g] > 0;
  if (rows_out <= 0 || rows_out > len) rows_out = len;
  if (len == 256 && tw256 && sign < 0 && real_in && rows_out == 129 && ld % 32 == 0 && all_valid) {
    hipLaunchKernelGGL(k_fft_cols256_real2, dim3(ld / 32, n_channels), dim3(256), 0, st,
                       reinterpret_cast<const float*>(in), out, ld, in_cstride, out_cstride, tw_n, tw256,
                       sums, inv_n, segs);
    GCWT_LAUNCH_CHECK();
    return hipSuccess;
  }
  if ((len == 512 || len == 1024) && tw256 && tw4096 && !(sign > 0 && real_in)) {
    const bool big = len == 1024;
    hipError_t e;
    if (sign < 0 && real_in && all_valid) {      // two real subsequences per FFT256
      if (big)
        hipLaunchKernelGGL(k_fft_colsq_real2<2>, dim3(ld / 16, n_channels), dim3(256), 0, st,
                           reinterpret_cast<const float*>(in), out, ld, in_cstride, out_cstride, tw_n, tw4096,
                           tw256, sums, inv_n, segs, rows_out);
      else
        hipLaunchKernelGGL(k_fft_colsq_real2<1>, dim3(ld / 16, n_channels), dim3(256), 0, st,
                           reinterpret_cast<const float*>(in), out, ld, in_cstride, out_cstride, tw_n, tw4096,
                           tw256, sums, inv_n, segs, rows_out);
      e = hipGetLastError();
    } else if (sign < 0 && real_in)
      e = big ? launch_colsq<-1, true, 2>(in, out, ld, in_cstride, out_cstride, tw_n, tw4096, tw256, sums, inv_n, segs, n_channels, rows_out, st)
              : launch_colsq<-1, true, 1>(in, out, ld, in_cstride, out_cstride, tw_n, tw4096, tw256, sums, inv_n, segs, n_channels, rows_out, st);
    else if (sign < 0)
      e = big ? launch_colsq<-1, false, 2>(in, out, ld, in_cstride, out_cstride, tw_n, tw4096, tw256, sums, inv_n, segs, n_channels, rows_out, st)
              : launch_colsq<-1, false, 1>(in, out, ld, in_cstride, out_cstride, tw_n, tw4096, tw256, sums, inv_n, segs, n_channels, rows_out, st);
    else
      e = big ? launch_colsq<1, false, 2>(in, out, ld, in_cstride, out_cstride, tw_n, tw4096, tw256, sums, inv_n, segs, n_channels, rows_out, st)
              : launch_colsq<1, false, 1>(in, out, ld, in_cstride, out_cstride, tw_n, tw4096, tw256, sums, inv_n, segs, n_channels, rows_out, st);
    return e;
  }
  if (len == 256 && tw256) {
    dim3 grid(ld / 16, n_channels), block(256);
    if (sign < 0 && real_in)
      hipLaunchKernelGGL((k_fft_cols256<-1, true>), grid, block, 0, st, in, out, ld, in_cstride, out_cstride, tw_n, tw256, sums, inv_n, segs, rows_out);
    else if (sign < 0)
      hipLaunchKernelGGL((k_fft_cols256<-1, false>), grid, block, 0, st, in, out, ld, in_cstride, out_cstride, tw_n, tw256, sums, inv_n, segs, rows_out);
    else if (real_in)
      hipLaunchKernelGGL((k_fft_cols256<1, true>), grid, block, 0, st, in, out, ld, in_cstride, out_cstride, tw_n, tw256, sums, inv_n, segs, rows_out);
    else
      hipLaunchKernelGGL((k_fft_cols256<1, false>), grid, block, 0, st, in, out, ld, in_cstride, out_cstride, tw_n, tw256, sums, inv_n, segs, rows_out);
    GCWT_LAUNCH_CHECK();
    return hipSuccess;
  }
  const size_t lds = (size_t)len * 16 * sizeof(cf);
  dim3 grid(ld / 16, n_channels), block(256);
  const int l2 = ilog2(len);
#define GCWT_COLS(S, RI)                                                                          \
  {                                                                                               \
    if (lds > 48 * 1024) {                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)k_fft_cols<S, RI>,                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
      if (e != hipSuccess) return e;                                                              \
    }                                                                                             \
    hipLaunchKernelGGL((k_fft_cols<S, RI>), grid, block, lds, st, in, out, len, l2, ld,           \
                       in_cstride, out_cstride, tw_n, tw4096, sums, inv_n, segs);                        \
  }
  if (sign < 0 && real_in) GCWT_COLS(-1, true)
  else if (sign < 0) GCWT_COLS(-1, false)
  else if (real_in) GCWT_COLS(1, true)
  else GCWT_COLS(1, false)
#undef GCWT_COLS
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_fft_rows(int sign, const cf* in, cf* out, int len, int64_t n_rows, int64_t in_ld,
                           int64_t out_ld, int64_t in_cstride, int64_t out_cstride, int64_t tw_n,
                           const cf* tw4096, const cf* tw256, float scale, int n_channels,
                           hipStream_t st, int out_len, int mirror, RowTaper taper) {
  const int l2 = ilog2(len);
  if (mirror && !(len == kRowLenDev && tw256)) return hipErrorInvalidValue;   // fast 4096-point rows only
  const int rows = kRowLenDev / len;
  dim3 grid((unsigned)((n_rows + rows - 1) / rows), n_channels), block(256);
  if (len >= 256 && tw256) {
    if (sign < 0)
      hipLaunchKernelGGL((k_fft_rows_fast<-1>), grid, block, 0, st, in, out, l2 - 8, in_ld, out_ld,
                         in_cstride, out_cstride, tw_n, tw4096, tw256, scale, (int)n_rows,
                         out_len > 0 ? out_len : len, mirror, taper);
    else
      hipLaunchKernelGGL((k_fft_rows_fast<1>), grid, block, 0, st, in, out, l2 - 8, in_ld, out_ld,
                         in_cstride, out_cstride, tw_n, tw4096, tw256, scale, (int)n_rows,
                         out_len > 0 ? out_len : len, mirror, taper);
    GCWT_LAUNCH_CHECK();
    return hipSuccess;
  }
  if (sign < 0)
    hipLaunchKernelGGL((k_fft_rows<-1>), grid, block, 0, st, in, out, len, l2, in_ld, out_ld,
                       in_cstride, out_cstride, tw_n, tw4096, scale, (int)n_rows, taper);
  else
    hipLaunchKernelGGL((k_fft_rows<1>), grid, block, 0, st, in, out, len, l2, in_ld, out_ld,
                       in_cstride, out_cstride, tw_n, tw4096, scale, (int)n_rows, taper);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_fullband_rows(const cf* x, const FullbandSet& set, int p1, int64_t x_cstride, int64_t z_cstride,
                                const cf* twt, const cf* tw256, int n_slots, hipStream_t st, int group) {
  if (p1 < 1 || !twt || !tw256 || set.n < 1 || set.n > kFullbandSet) return hipErrorInvalidValue;
  for (int k = 0; k < set.n; ++k)
    if (!set.h[k] || !set.z[k]) return hipErrorInvalidValue;
  if ((int64_t)n_slots * p1 > 0x7fffffff) return hipErrorInvalidValue;
  const int64_t p = (int64_t)p1 * kRowLenDev;
  if (group < 1 || group > p1 || p1 % group) group = p1 < 32 ? p1 : 32;   // p1 is a power of two
  hipLaunchKernelGGL(k_fullband_rows, dim3((unsigned)(n_slots * p1)), dim3(256), 0, st, x, set, x_cstride, z_cstride,
                     p1 > 1 ? p : 0, twt, tw256, n_slots, group);
  return hipGetLastError();
}

hipError_t launch_fullband4(int mode, const cf* x, const cf* const* h, const int32_t* scales, int n, float* out,
                            int64_t x_cstride, const cf* twt, const cf* tw256, int n_scales, int64_t row_len,
                            const SegOut& seg, int n_slots, hipStream_t st) {
  if (n < 1 || n > kFullband4Scales || !twt || !tw256 || n_slots < 1) return hipErrorInvalidValue;
  Fullband4Set set{};
  set.n = n;
  for (int k = 0; k < n; ++k) {
    if (!h[k]) return hipErrorInvalidValue;
    set.h[k] = h[k];
    set.scale[k] = scales[k];
  }
  // few slots: the scales of the set go to workgroups of their own
  const int split = n_slots >= 2048 ? 1 : std::min(n, std::max(1, 2048 / n_slots));
  const dim3 grid((unsigned)n_slots, (unsigned)split), block(512);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_fullband4<GCWT_OUT_AMPLITUDE_F32>), grid, block, 0, st, x, set, out, x_cstride, twt, tw256, n_scales, row_len, seg);
  else if (mode == GCWT_OUT_POWER_F32)
    hipLaunchKernelGGL((k_fullband4<GCWT_OUT_POWER_F32>), grid, block, 0, st, x, set, out, x_cstride, twt, tw256, n_scales, row_len, seg);
  else
    hipLaunchKernelGGL((k_fullband4<GCWT_OUT_COMPLEX_C64>), grid, block, 0, st, x, set, out, x_cstride, twt, tw256, n_scales, row_len, seg);
  return hipGetLastError();
}

bool fullband_cols_fused(int p1) { return p1 == 256 || p1 == 512 || p1 == 1024; }

template <int MODE>
static void launch_fullband_cols_mode(const cf* z, float* out, int p1, int64_t z_cstride, const cf* tw4096,
                                      const cf* tw256, int scale, int n_scales, int64_t row_len,
                                      const SegOut& seg, int n_slots, hipStream_t st) {
  const dim3 grid(kRowLenDev / 16, n_slots), block(256);
  if (p1 == 256)
    hipLaunchKernelGGL((k_fullband_cols256<MODE>), grid, block, 0, st, z, out, kRowLenDev, z_cstride, tw256,
                       scale, n_scales, row_len, seg);
  else if (p1 == 512)
    hipLaunchKernelGGL((k_fullband_colsq<MODE, 1>), grid, block, 0, st, z, out, kRowLenDev, z_cstride, tw4096,
                       tw256, scale, n_scales, row_len, seg);
  else
    hipLaunchKernelGGL((k_fullband_colsq<MODE, 2>), grid, block, 0, st, z, out, kRowLenDev, z_cstride, tw4096,
                       tw256, scale, n_scales, row_len, seg);
}

hipError_t launch_fullband_cols(int mode, const cf* z, float* out, int p1, int64_t z_cstride, const cf* tw4096,
                                const cf* tw256, int scale, int n_scales, int64_t row_len, const SegOut& seg,
                                int n_slots, hipStream_t st) {
  if (!fullband_cols_fused(p1) || !tw4096 || !tw256) return hipErrorInvalidValue;
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    launch_fullband_cols_mode<GCWT_OUT_AMPLITUDE_F32>(z, out, p1, z_cstride, tw4096, tw256, scale, n_scales, row_len, seg, n_slots, st);
  else if (mode == GCWT_OUT_POWER_F32)
    launch_fullband_cols_mode<GCWT_OUT_POWER_F32>(z, out, p1, z_cstride, tw4096, tw256, scale, n_scales, row_len, seg, n_slots, st);
  else
    launch_fullband_cols_mode<GCWT_OUT_COMPLEX_C64>(z, out, p1, z_cstride, tw4096, tw256, scale, n_scales, row_len, seg, n_slots, st);
  return hipGetLastError();
}

hipError_t launch_bc_scales(int mode, const cf* xb, float* out, const cf* h, const int32_t* rows,
                            int n_group_scales, const cf* twt, const cf* tw256, const BcBlocks& bl, int blk0,
                            int nblk, int n_scales, int64_t col0, int64_t row_len, hipStream_t st,
                            const unsigned char* mask) {
  if (nblk <= 0 || n_group_scales <= 0) return hipSuccess;
  if (bl.n_epochs < 1 || bl.n_epochs > kSegBatch || bl.hop < 1 || bl.back < 0 || bl.hop + bl.back > kRowLenDev ||
      blk0 < 0 || blk0 + nblk > bl.blk_first[bl.n_epochs] || (int64_t)nblk * bl.n_channels > 0x7fffffff)
    return hipErrorInvalidValue;
  const dim3 grid((unsigned)(nblk * bl.n_channels)), block(256);
#define GCWT_BC(M)                                                                                        \
  hipLaunchKernelGGL((k_bc_scales<M>), grid, block, 0, st, xb, out, h, rows, n_group_scales, twt, tw256, \
                     bl, blk0, n_scales, col0, row_len, mask)
  if (mode == GCWT_OUT_AMPLITUDE_F32) GCWT_BC(GCWT_OUT_AMPLITUDE_F32);
  else if (mode == GCWT_OUT_POWER_F32) GCWT_BC(GCWT_OUT_POWER_F32);
  else GCWT_BC(GCWT_OUT_COMPLEX_C64);
#undef GCWT_BC
  return hipGetLastError();
}

hipError_t launch_block_fft(const cf* xr, cf* xb, int64_t m, int hop, int halo, int blk_lo, int nblk,
                            int64_t xr_cstride, int64_t xb_cstride, const cf* tw256, float scale,
                            int n_channels, hipStream_t st) {
  dim3 grid((nblk + 15) / 16, n_channels), block(256);
  hipLaunchKernelGGL(k_block_fft, grid, block, 0, st, xr, xb, m - 1, hop, halo, blk_lo, nblk, xr_cstride,
                     xb_cstride, tw256, scale);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_synth(int mode, const SynthArgs& a, int n_items, int n_channels, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  dim3 grid(n_items, n_channels), block(256);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth<GCWT_OUT_AMPLITUDE_F32>), grid, block, 0, st, a);
  else if (mode == GCWT_OUT_POWER_F32)
    hipLaunchKernelGGL((k_synth<GCWT_OUT_POWER_F32>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((k_synth<GCWT_OUT_COMPLEX_C64>), grid, block, 0, st, a);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_direct(int mode, const float* x, float* out, const cf* psi, const DirectScale* sc,
                         int n_direct, const double* sums, double inv_n, int64_t n_samples,
                         int n_scales, const DirectEpochs& eps, int n_epochs, int64_t col0,
                         int64_t row_len, int64_t max_len, const cf* tail, hipStream_t st,
                         const unsigned char* mask) {
  int64_t longest = 0;
  for (int e = 0; e < n_epochs; ++e) longest = std::max(longest, eps.g_hi[e] - eps.g_lo[e]);
  if (n_direct == 0 || n_epochs == 0 || longest <= 0) return hipSuccess;
  // samples either side of a tile that its outputs reach: half the longest kernel, and the
  // zero-padded tail of the last group of 8 taps
  // (front zeros included: top + front + 1 <= halo and L - top + 6 <= halo)
  const int halo = (int)(((max_len / 2 + 9) + 7) & ~7);
  const int elem = mode == GCWT_OUT_COMPLEX_C64 ? 2 : 1;
  const size_t lds = sizeof(float) * (size_t)((kDirectTile + 2 * halo) / 8 * 12 + 4 * 544 * elem + (kDirectTile + 2 * halo));
  dim3 grid((unsigned)((longest + kDirectTile - 1) / kDirectTile), 1, eps.n_channels * n_epochs), block(256);
#define GCWT_DIRECT(M)                                                                       \
  hipLaunchKernelGGL((k_direct<M>), grid, block, lds, st, x, out, psi, sc, n_direct, sums,   \
                     inv_n, n_samples, n_scales, eps, col0, row_len, halo, tail, mask)
  if (mode == GCWT_OUT_AMPLITUDE_F32) GCWT_DIRECT(GCWT_OUT_AMPLITUDE_F32);
  else if (mode == GCWT_OUT_POWER_F32) GCWT_DIRECT(GCWT_OUT_POWER_F32);
  else GCWT_DIRECT(GCWT_OUT_COMPLEX_C64);
#undef GCWT_DIRECT
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_level_small(const cf* x, cf* xr, int n1, int q, int64_t row_stride,
                              int64_t x_cstride, int64_t xr_cstride, const cf* tw4096,
                              int n_channels, hipStream_t st, RowTaper taper) {
  const size_t lds = (size_t)n1 * q * sizeof(cf);
  static bool attr_done[64] = {};            // per device: one process may drive several
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_done[dev_ & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_level_small,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_level_small, dim3(n_channels), dim3(256), lds, st, x, xr, n1, ilog2(n1), q,
                     row_stride, x_cstride, xr_cstride, tw4096, taper);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_cmul_inplace(cf* a, const cf* b, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_cmul_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_chirp_kernel(cf* b, int64_t N, int64_t P, hipStream_t st) {
  hipLaunchKernelGGL(k_chirp_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, b, N, P);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_chirp_load(const float* v, int is_complex, int conj_in, int64_t n_valid, int64_t N,
                             int64_t P, const double* sums, double inv_n, cf* a, hipStream_t st) {
  hipLaunchKernelGGL(k_chirp_load, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, v, is_complex,
                     conj_in, n_valid, N, P, sums, inv_n, a);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_chirp_analytic_mask(cf* y, int64_t N, int64_t P, float scale, hipStream_t st) {
  hipLaunchKernelGGL(k_chirp_analytic_mask, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, y, N,
                     P, scale);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_chirp_store(const cf* y, cf* out, int64_t count, int64_t N, float scale,
                              int conj_out, const double* sums, double inv_n, hipStream_t st) {
  hipLaunchKernelGGL(k_chirp_store, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, y, out,
                     count, N, scale, conj_out, sums, inv_n);
  GCWT_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_zero_range(float* out, int64_t row_len_floats, int64_t n_rows, int64_t start,
                             int64_t len, hipStream_t st) {
  if (len <= 0) return hipSuccess;
  // n_rows can exceed 65535: split over calls
  for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
    const int64_t nr = std::min<int64_t>(65535, n_rows - r0);
    dim3 grid((unsigned)((len + 255) / 256), (unsigned)nr), block(256);
    hipLaunchKernelGGL(k_zero_range, grid, block, 0, st, out + r0 * row_len_floats, row_len_floats,
                       start, len);
    GCWT_LAUNCH_CHECK();
  }
  return hipSuccess;
}

}  // namespace gcwt
