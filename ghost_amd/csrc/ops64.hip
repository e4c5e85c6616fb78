// ops64.hip -- the operator layer in float64 (include/ghostcwt.h: gcwt_fastconv_f64, gcwt_dft_f64): what the
// reference's fastconv_* / chirpz_dft / analytic_signal_* return -- complex128 from float64 FFTs
// (ghost/sigtools/convolution.py:68-87, fourier.py:9-52, analytic.py:22-112) -- for callers who compare element by
// element (np.allclose at a result's zero crossings asks for 1e-8 of its peak, which no float32 transform gives).
// A Stockham autosort FFT in global memory, radix 4 (+ one radix-2 stage for odd log2 lengths), twiddles as products
// of two float64 table entries (fwd64.hip's tables); arbitrary lengths through Bluestein's chirp-z identity with the
// chirp's phase reduced in integers.  HBM-bound and simple: five or six passes over the array per transform.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace gcwt {

namespace {

typedef double2 cd;
__device__ __forceinline__ cd cmul64(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cd cadd64(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd csub64(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }

// exp(sign 2 pi i num / 2^lg): tables tw_hi[a] = exp(-2 pi i a / 4096), tw_lo[b] = exp(-2 pi i b / 2^24)
__device__ __forceinline__ cd phase64(const cd* __restrict__ tw, int64_t num, int lg, int sign) {
  const int64_t m = (num << (24 - lg)) & (((int64_t)1 << 24) - 1);
  cd w = cmul64(tw[m >> 12], tw[4096 + (m & 4095)]);
  if (sign > 0) w.y = -w.y;
  return w;
}

// One Stockham stage of radix R over a batch: n = 2^lg points, ns = size of the transforms done so far.
//   j < n / R, k = j mod ns: u[r] = x[j + r n / R] W^(r k), W = exp(sign 2 pi i / (R ns)); y[(j - k) R + k + r ns] = DFT_R(u)[r]
template <int R>
__global__ void __launch_bounds__(256) k_stage64(const cd* __restrict__ x, cd* __restrict__ y, int lg, int lg_ns, int sign,
                                                 const cd* __restrict__ tw) {
  const int64_t n = (int64_t)1 << lg, nr = n / R;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= nr) return;
  const cd* xb = x + (int64_t)blockIdx.y * n;
  cd* yb = y + (int64_t)blockIdx.y * n;
  const int64_t ns = (int64_t)1 << lg_ns, k = j & (ns - 1);
  const int lg_w = lg_ns + (R == 4 ? 2 : 1);                    // W = exp(sign 2 pi i / 2^lg_w)
  cd u[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    u[r] = xb[j + r * nr];
    if (r > 0 && k > 0) u[r] = cmul64(u[r], phase64(tw, r * k, lg_w, sign));
  }
  if (R == 2) {
    const cd a = cadd64(u[0], u[1]), b = csub64(u[0], u[1]);
    u[0] = a; u[1] = b;
  } else {
    const cd a = cadd64(u[0], u[2]), b = csub64(u[0], u[2]), c = cadd64(u[1], u[3]), d = csub64(u[1], u[3]);
    // times sign * i: forward (sign < 0) -i d, inverse +i d
    const cd id = sign < 0 ? make_double2(d.y, -d.x) : make_double2(-d.y, d.x);
    u[0] = cadd64(a, c); u[2] = csub64(a, c); u[1] = cadd64(b, id); u[3] = csub64(b, id);
  }
  const int64_t j0 = (j - k) * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) yb[j0 + r * ns] = u[r];
}

__global__ void __launch_bounds__(256) k_mul64(cd* __restrict__ a, const cd* __restrict__ b, int64_t n, double scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const cd p = cmul64(a[i], b[i]); a[i] = make_double2(p.x * scale, p.y * scale); }
}

// Bluestein: c[n] = exp(sign i pi n^2 / N), the phase n^2 mod 2 N in integers
__device__ __forceinline__ cd chirp64(int64_t n, int64_t N, int sign) {
  const int64_t q = (n * n) % (2 * N);
  double s, c;
  sincospi((double)q / (double)N, &s, &c);
  return make_double2(c, sign < 0 ? -s : s);
}
// a[n] = x[n] c[n] (n < N), 0 beyond; x real or complex
__global__ void __launch_bounds__(256) k_chirp_in64(const double* __restrict__ x, int is_complex, cd* __restrict__ a,
                                                    int64_t N, int64_t L, int sign) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  if (i >= N) { a[i] = make_double2(0.0, 0.0); return; }
  const cd v = is_complex ? make_double2(x[2 * i], x[2 * i + 1]) : make_double2(x[i], 0.0);
  a[i] = cmul64(v, chirp64(i, N, sign));
}
// b[n] = conj(c[n]) for |n| < N, wrapped onto L points
__global__ void __launch_bounds__(256) k_chirp_kernel64(cd* __restrict__ b, int64_t N, int64_t L, int sign) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  const int64_t n = i < N ? i : (L - i < N ? L - i : -1);
  b[i] = n < 0 ? make_double2(0.0, 0.0) : chirp64(n, N, -sign);
}
// out[k] = conv[k] c[k] scale
__global__ void __launch_bounds__(256) k_chirp_out64(const cd* __restrict__ conv, double* __restrict__ out, int64_t N, int sign,
                                                     double scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const cd v = cmul64(conv[i], chirp64(i, N, sign));
  out[2 * i] = v.x * scale;
  out[2 * i + 1] = v.y * scale;
}
__global__ void __launch_bounds__(256) k_load64(const double* __restrict__ x, int is_complex, cd* __restrict__ a, int64_t n, int64_t L) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  a[i] = i < n ? (is_complex ? make_double2(x[2 * i], x[2 * i + 1]) : make_double2(x[i], 0.0)) : make_double2(0.0, 0.0);
}
__global__ void __launch_bounds__(256) k_store64(const cd* __restrict__ a, double* __restrict__ out, int64_t first, int64_t count,
                                                 double scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  out[2 * i] = a[first + i].x * scale;
  out[2 * i + 1] = a[first + i].y * scale;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }
inline int ilog2(int64_t v) { int l = 0; while (((int64_t)1 << l) < v) ++l; return l; }

struct Dev64 {
  std::mutex mu;
  cd* tw = nullptr;         // [4096 + 4096]
  int device = -1;
};
Dev64& dev64() { static Dev64 d; return d; }

hipError_t tables(cd** out) {
  Dev64& d = dev64();
  int cur = 0;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return e;
  if (d.tw && d.device != cur) { (void)hipFree(d.tw); d.tw = nullptr; }
  if (!d.tw) {
    std::vector<double2> h(8192);
    fwd64_fill_tables(h.data());
    if ((e = hipMalloc((void**)&d.tw, sizeof(cd) * 8192)) != hipSuccess) return e;
    if ((e = hipMemcpy(d.tw, h.data(), sizeof(cd) * 8192, hipMemcpyHostToDevice)) != hipSuccess) return e;
    d.device = cur;
  }
  *out = d.tw;
  return hipSuccess;
}

// 2^lg-point transform of buffer a (work: b); the result's buffer is returned in *res
hipError_t fft64(cd* a, cd* b, int lg, int sign, const cd* tw, hipStream_t st, cd** res) {
  const int64_t n = (int64_t)1 << lg;
  cd *src = a, *dst = b;
  int lg_ns = 0;
  if (lg & 1) {
    hipLaunchKernelGGL(k_stage64<2>, grid1(n / 2), dim3(256), 0, st, src, dst, lg, lg_ns, sign, tw);
    std::swap(src, dst);
    lg_ns += 1;
  }
  for (; lg_ns < lg; lg_ns += 2) {
    hipLaunchKernelGGL(k_stage64<4>, grid1(n / 4), dim3(256), 0, st, src, dst, lg, lg_ns, sign, tw);
    std::swap(src, dst);
  }
  *res = src;
  return hipGetLastError();
}

struct Scratch {
  void* p[4] = {nullptr, nullptr, nullptr, nullptr};
  ~Scratch() { for (void* q : p) if (q) (void)hipFree(q); }
  hipError_t alloc(int i, size_t bytes) { return hipMalloc(&p[i], bytes ? bytes : 1); }
};

}  // namespace

// DFT (inverse: normalised) of n points, any n <= 2^23; x, out on the host (out: (re, im) pairs)
hipError_t dft_f64(const double* x, int64_t n, int is_complex, int inverse, double* out) {
  std::lock_guard<std::mutex> lock(dev64().mu);
  cd* tw = nullptr;
  hipError_t e = tables(&tw);
  if (e != hipSuccess) return e;
  const int sign = inverse ? +1 : -1;
  const double scale = inverse ? 1.0 / (double)n : 1.0;
  const bool pow2 = (n & (n - 1)) == 0;
  const int64_t L = pow2 ? n : (int64_t)1 << ilog2(2 * n - 1);
  const int lg = ilog2(L);
  if (lg > 24) return hipErrorInvalidValue;
  Scratch s;
  const size_t in_bytes = sizeof(double) * (size_t)n * (is_complex ? 2 : 1);
  if ((e = s.alloc(0, in_bytes)) != hipSuccess || (e = s.alloc(1, sizeof(cd) * (size_t)L)) != hipSuccess ||
      (e = s.alloc(2, sizeof(cd) * (size_t)L)) != hipSuccess || (e = s.alloc(3, sizeof(cd) * (size_t)std::max<int64_t>(L, n))) != hipSuccess)
    return e;
  double* d_in = (double*)s.p[0];
  cd *a = (cd*)s.p[1], *b = (cd*)s.p[2], *c = (cd*)s.p[3];
  if ((e = hipMemcpy(d_in, x, in_bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
  cd* res = nullptr;
  if (pow2) {
    hipLaunchKernelGGL(k_load64, grid1(L), dim3(256), 0, nullptr, d_in, is_complex, a, n, L);
    if ((e = fft64(a, b, lg, sign, tw, nullptr, &res)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_store64, grid1(n), dim3(256), 0, nullptr, res, (double*)c, (int64_t)0, n, scale);
  } else {
    // X[k] = c[k] sum_n (x[n] c[n]) conj(c)[k - n], c[n] = exp(sign i pi n^2 / N): a circular convolution of length L
    cd *fa = nullptr, *fb = nullptr, *fc = nullptr;
    hipLaunchKernelGGL(k_chirp_in64, grid1(L), dim3(256), 0, nullptr, d_in, is_complex, a, n, L, sign);
    if ((e = fft64(a, b, lg, -1, tw, nullptr, &fa)) != hipSuccess) return e;
    cd* other = fa == a ? b : a;                                  // the buffer the spectrum of a does not sit in
    hipLaunchKernelGGL(k_chirp_kernel64, grid1(L), dim3(256), 0, nullptr, c, n, L, sign);
    Scratch s2;
    if ((e = s2.alloc(0, sizeof(cd) * (size_t)L)) != hipSuccess) return e;
    if ((e = fft64(c, (cd*)s2.p[0], lg, -1, tw, nullptr, &fb)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_mul64, grid1(L), dim3(256), 0, nullptr, fa, fb, L, 1.0 / (double)L);
    if ((e = fft64(fa, other, lg, +1, tw, nullptr, &fc)) != hipSuccess) return e;
    double* d_out = (double*)(fc == a ? b : a);                   // 2 n doubles <= L complex
    hipLaunchKernelGGL(k_chirp_out64, grid1(n), dim3(256), 0, nullptr, fc, d_out, n, sign, scale);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    return hipMemcpy(out, d_out, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  return hipMemcpy(out, c, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost);
}

// Results beyond one 2^24-point transform: overlap-add over chunks of the signal, the kernel's spectrum made once --
// what convolution.py:70-77 does on the host (chunks of chunk_size samples, each chunk's full result added where it
// belongs).  A chunk's transform is 2^22 points (four times the kernel for longer kernels, 2^24 at most), its
// Lc - m + 1 signal samples give Lc result samples of which the part inside [first, first + count) is added into out
// on the host, in chunk order: every sum has the same order every run.
static hipError_t fastconv_f64_chunked(const double* signal, int64_t n, int signal_is_complex, const double* kernel, int64_t m,
                                       int kernel_is_complex, int64_t first, int64_t count, double* out, const cd* tw) {
  if (m > ((int64_t)1 << 23)) return hipErrorInvalidValue;
  const int lg = std::min(24, std::max(22, ilog2(4 * m)));
  const int64_t L = (int64_t)1 << lg, B = L - m + 1;           // B >= m: a result sample has two chunks' parts at most
  const int sw = signal_is_complex ? 2 : 1, kw = kernel_is_complex ? 2 : 1;
  hipError_t e;
  Scratch s, s2;
  if ((e = s.alloc(0, sizeof(double) * (size_t)std::max(B * sw, m * kw))) != hipSuccess || (e = s.alloc(1, sizeof(cd) * (size_t)L)) != hipSuccess ||
      (e = s.alloc(2, sizeof(cd) * (size_t)L)) != hipSuccess || (e = s.alloc(3, sizeof(cd) * (size_t)L)) != hipSuccess ||
      (e = s2.alloc(0, sizeof(cd) * (size_t)L)) != hipSuccess)
    return e;
  double* d_in = (double*)s.p[0];
  cd *a = (cd*)s.p[1], *b = (cd*)s.p[2], *c = (cd*)s.p[3], *d = (cd*)s2.p[0];
  cd* fk = nullptr;
  if ((e = hipMemcpy(d_in, kernel, sizeof(double) * (size_t)(m * kw), hipMemcpyHostToDevice)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_load64, grid1(L), dim3(256), 0, nullptr, d_in, kernel_is_complex, c, m, L);
  if ((e = fft64(c, d, lg, -1, tw, nullptr, &fk)) != hipSuccess) return e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  std::vector<double> part((size_t)2 * (size_t)L);
  std::memset(out, 0, sizeof(double) * 2 * (size_t)count);
  for (int64_t s0 = 0; s0 < n; s0 += B) {
    const int64_t nb = std::min(B, n - s0), len = nb + m - 1;            // the chunk's result: [s0, s0 + len) of the full one
    const int64_t lo = std::max(s0, first), hi = std::min(s0 + len, first + count);
    if (hi <= lo) continue;
    if ((e = hipMemcpy(d_in, signal + s0 * sw, sizeof(double) * (size_t)(nb * sw), hipMemcpyHostToDevice)) != hipSuccess) return e;
    cd *fa = nullptr, *fc = nullptr;
    hipLaunchKernelGGL(k_load64, grid1(L), dim3(256), 0, nullptr, d_in, signal_is_complex, a, nb, L);
    if ((e = fft64(a, b, lg, -1, tw, nullptr, &fa)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_mul64, grid1(L), dim3(256), 0, nullptr, fa, fk, L, 1.0 / (double)L);
    if ((e = fft64(fa, fa == a ? b : a, lg, +1, tw, nullptr, &fc)) != hipSuccess) return e;
    double* d_out = (double*)(fc == a ? b : a);
    hipLaunchKernelGGL(k_store64, grid1(hi - lo), dim3(256), 0, nullptr, fc, d_out, lo - s0, hi - lo, 1.0);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    if ((e = hipMemcpy(part.data(), d_out, sizeof(double) * 2 * (size_t)(hi - lo), hipMemcpyDeviceToHost)) != hipSuccess) return e;
    double* dst = out + 2 * (lo - first);
    for (int64_t i = 0; i < 2 * (hi - lo); ++i) dst[i] += part[(size_t)i];
  }
  return hipSuccess;
}

// Linear convolution of signal (n) and kernel (m), both real or complex float64 on the host; samples
// [first, first + count) of the full result to out ((re, im) pairs).  One FFT of 2^k >= n + m - 1 points up to 2^24;
// longer results by overlap-add (above).
hipError_t fastconv_f64(const double* signal, int64_t n, int signal_is_complex, const double* kernel, int64_t m,
                        int kernel_is_complex, int64_t first, int64_t count, double* out) {
  std::lock_guard<std::mutex> lock(dev64().mu);
  cd* tw = nullptr;
  hipError_t e = tables(&tw);
  if (e != hipSuccess) return e;
  if (n + m - 1 > ((int64_t)1 << 24))
    return fastconv_f64_chunked(signal, n, signal_is_complex, kernel, m, kernel_is_complex, first, count, out, tw);
  const int lg = ilog2(n + m - 1);
  if (lg > 24) return hipErrorInvalidValue;
  const int64_t L = (int64_t)1 << lg;
  Scratch s;
  const size_t sb = sizeof(double) * (size_t)n * (signal_is_complex ? 2 : 1), kb = sizeof(double) * (size_t)m * (kernel_is_complex ? 2 : 1);
  if ((e = s.alloc(0, std::max(sb, kb))) != hipSuccess || (e = s.alloc(1, sizeof(cd) * (size_t)L)) != hipSuccess ||
      (e = s.alloc(2, sizeof(cd) * (size_t)L)) != hipSuccess || (e = s.alloc(3, sizeof(cd) * (size_t)L)) != hipSuccess)
    return e;
  Scratch s2;
  if ((e = s2.alloc(0, sizeof(cd) * (size_t)L)) != hipSuccess) return e;
  double* d_in = (double*)s.p[0];
  cd *a = (cd*)s.p[1], *b = (cd*)s.p[2], *c = (cd*)s.p[3], *d = (cd*)s2.p[0];
  cd *fa = nullptr, *fk = nullptr, *fc = nullptr;
  if ((e = hipMemcpy(d_in, signal, sb, hipMemcpyHostToDevice)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_load64, grid1(L), dim3(256), 0, nullptr, d_in, signal_is_complex, a, n, L);
  if ((e = fft64(a, b, lg, -1, tw, nullptr, &fa)) != hipSuccess) return e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  if ((e = hipMemcpy(d_in, kernel, kb, hipMemcpyHostToDevice)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_load64, grid1(L), dim3(256), 0, nullptr, d_in, kernel_is_complex, c, m, L);
  if ((e = fft64(c, d, lg, -1, tw, nullptr, &fk)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_mul64, grid1(L), dim3(256), 0, nullptr, fa, fk, L, 1.0 / (double)L);
  cd* other = fa == a ? b : a;
  if ((e = fft64(fa, other, lg, +1, tw, nullptr, &fc)) != hipSuccess) return e;
  double* d_out = (double*)(fk == c ? d : c);                    // count <= L complex
  hipLaunchKernelGGL(k_store64, grid1(count), dim3(256), 0, nullptr, fc, d_out, first, count, 1.0);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
  return hipMemcpy(out, d_out, sizeof(double) * 2 * (size_t)count, hipMemcpyDeviceToHost);
}

}  // namespace gcwt
