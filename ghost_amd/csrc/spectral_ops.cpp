// spectral_ops.cpp -- one-shot spectral operators beside the transform: arbitrary-length
// DFT (Bluestein / chirp-z) and the analytic signal, built on the same power-of-two device
// FFTs the transform uses.  Reference: ghost/sigtools/fourier.py:9-52 (chirpz_dft),
// ghost/sigtools/analytic.py:22-112 (analytic_signal_fftw).
#include <hip/hip_runtime.h>

#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "../../include/ghostcwt.h"
#include "kernels.h"
#include "planner.h"

using namespace gcwt;

int gcwt_internal_set_error(int code, const char* msg);   // api.cpp (C++ linkage)

namespace {

int fail(int code, const std::string& m) { return gcwt_internal_set_error(code, m.c_str()); }

// Circular convolution engine of one power-of-two length P = P1 * 4096 (<= 2^22): owns the
// stream, the twiddle tables and three P-point work arrays.
struct ChirpEngine {
  int64_t N = 0, P = 0;
  int P1 = 0;
  hipStream_t st = nullptr;
  float2 *a = nullptr, *bspec = nullptr, *out = nullptr, *tw4096 = nullptr, *tw256 = nullptr;
  float* in = nullptr;
  double* sum = nullptr;   // [0] = sum of the input (or 0), read by the load/store kernels
  hipError_t err = hipSuccess;
  const char* where = "";

  ~ChirpEngine() {
    (void)hipFree(a); (void)hipFree(bspec); (void)hipFree(out); (void)hipFree(tw4096);
    (void)hipFree(tw256); (void)hipFree(in); (void)hipFree(sum);
    if (st) (void)hipStreamDestroy(st);
  }
  bool ok(hipError_t e, const char* w) {
    if (e != hipSuccess && err == hipSuccess) { err = e; where = w; }
    return err == hipSuccess;
  }
#define CE(call) if (!ok((call), #call)) return false

  bool init(int64_t n_dft, size_t in_bytes, int64_t out_count) {
    N = n_dft;
    P = kRowLen;
    while (P < 2 * N - 1) P <<= 1;
    P1 = (int)(P / kRowLen);
    CE(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CE(hipMalloc((void**)&a, sizeof(float2) * P));
    CE(hipMalloc((void**)&bspec, sizeof(float2) * P));
    CE(hipMalloc((void**)&out, sizeof(float2) * out_count));
    CE(hipMalloc((void**)&in, in_bytes));
    CE(hipMalloc((void**)&sum, sizeof(double) * channel_sum_doubles(1)));
    std::vector<float2> t4(kRowLen / 2), t2(256);
    for (int j = 0; j < kRowLen / 2; ++j) {
      const double x = -2.0 * M_PI * j / kRowLen;
      t4[j] = make_float2((float)std::cos(x), (float)std::sin(x));
    }
    for (int q = 0; q < 256; ++q) {
      const double x = 2.0 * M_PI * q / 256.0;
      t2[q] = make_float2((float)std::cos(x), (float)std::sin(x));
    }
    CE(hipMalloc((void**)&tw4096, sizeof(float2) * t4.size()));
    CE(hipMalloc((void**)&tw256, sizeof(float2) * 256));
    CE(hipMemcpyAsync(tw4096, t4.data(), sizeof(float2) * t4.size(), hipMemcpyHostToDevice, st));
    CE(hipMemcpyAsync(tw256, t2.data(), sizeof(float2) * 256, hipMemcpyHostToDevice, st));
    CE(hipMemsetAsync(sum, 0, sizeof(double) * channel_sum_doubles(1), st));
    CE(hipStreamSynchronize(st));   // the host vectors go out of scope
    // spectrum of the chirp kernel, in the forward FFT's k1-major order
    CE(launch_chirp_kernel(bspec, N, P, st));
    return forward(bspec);
  }
  bool forward(float2* v) {
    CE(launch_fft_cols(-1, false, v, v, P1, kRowLen, 0, 0, P1 > 1 ? P : 0, tw4096, tw256, sum, 0.0,
                       0, 1, st));
    CE(launch_fft_rows(-1, v, v, kRowLen, P1, kRowLen, kRowLen, 0, 0, 0, tw4096, tw256, 1.0f, 1, st));
    return true;
  }
  // a <- circular convolution of a with the chirp kernel (unnormalised: P times too large)
  bool convolve() {
    if (!forward(a)) return false;
    CE(launch_cmul_inplace(a, bspec, P, st));
    CE(launch_fft_rows(+1, a, a, kRowLen, P1, kRowLen, kRowLen, 0, 0, P1 > 1 ? P : 0, tw4096, tw256,
                       1.0f, 1, st));
    if (P1 > 1)
      CE(launch_fft_cols(+1, false, a, a, P1, kRowLen, 0, 0, 0, tw4096, tw256, sum, 0.0, 0, 1, st));
    return true;
  }
#undef CE
};

int check_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(GCWT_ERR_NO_DEVICE, "no HIP device: libghostcwt has no CPU path");
  if (device >= 0 && hipSetDevice(device) != hipSuccess)
    return fail(GCWT_ERR_HIP, "hipSetDevice failed");
  return GCWT_OK;
}

// nothing may unwind across the C ABI
template <typename F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(GCWT_ERR_NOMEM, "out of host memory");
  } catch (...) {
    return fail(GCWT_ERR_INVALID, "internal error");
  }
}

int engine_error(const ChirpEngine& e) {
  return fail(GCWT_ERR_HIP, std::string(e.where) + ": " + hipGetErrorString(e.err));
}

}  // namespace

static int dft_impl(const float* x, int64_t n, int is_complex, int inverse, float* out, int device) {
  if (!x || !out || n <= 0) return fail(GCWT_ERR_INVALID, "bad argument");
  if (2 * n - 1 > (int64_t)kRowLen * kMaxP1)
    return fail(GCWT_ERR_UNSUPPORTED, "DFT length exceeds 2^21");
  int rc = check_device(device);
  if (rc) return rc;
  ChirpEngine e;
  const size_t in_bytes = sizeof(float) * (size_t)n * (is_complex ? 2 : 1);
  if (!e.init(n, in_bytes, n)) return engine_error(e);
  const float scale = (float)(1.0 / (double)e.P / (inverse ? (double)n : 1.0));
  bool good = e.ok(hipMemcpyAsync(e.in, x, in_bytes, hipMemcpyHostToDevice, e.st), "copy in") &&
              e.ok(launch_chirp_load(e.in, is_complex, inverse, n, n, e.P, e.sum, 0.0, e.a, e.st),
                   "chirp_load") &&
              e.convolve() &&
              e.ok(launch_chirp_store(e.a, e.out, n, n, scale, inverse, e.sum, 0.0, e.st),
                   "chirp_store") &&
              e.ok(hipMemcpyAsync(out, e.out, sizeof(float2) * n, hipMemcpyDeviceToHost, e.st),
                   "copy out") &&
              e.ok(hipStreamSynchronize(e.st), "sync");
  return good ? GCWT_OK : engine_error(e);
}

static int analytic_impl(const float* signal, int64_t n, int64_t fft_length, float* out,
                         int device) {
  if (!signal || !out) return fail(GCWT_ERR_INVALID, "bad argument");
  if (n <= 0) return fail(GCWT_ERR_INVALID, "Cannot compute analytic signal on an empty array");
  if (fft_length == 0) fft_length = n;
  if (fft_length < n)
    return fail(GCWT_ERR_INVALID, "'fft_length' must be at least the length of the input data");
  const int64_t F = fft_length;
  if (2 * F - 1 > (int64_t)kRowLen * kMaxP1)
    return fail(GCWT_ERR_UNSUPPORTED, "fft_length exceeds 2^21");
  int rc = check_device(device);
  if (rc) return rc;
  ChirpEngine e;
  if (!e.init(F, sizeof(float) * (size_t)n, n)) return engine_error(e);
  // The mean over the F-point frame is taken out in fp64 first and added back at the end (a
  // constant is its own analytic signal), so a DC offset costs no fp32 precision.
  const double inv_f = 1.0 / (double)F;
  const double inv_p = 1.0 / (double)e.P;
  bool good =
      e.ok(hipMemcpyAsync(e.in, signal, sizeof(float) * n, hipMemcpyHostToDevice, e.st), "copy in") &&
      e.ok(launch_channel_sum(e.in, n, 1, e.sum, e.st), "channel_sum") &&
      e.ok(launch_chirp_load(e.in, 0, 0, n, F, e.P, e.sum, inv_f, e.a, e.st), "chirp_load") &&
      e.convolve() &&
      e.ok(launch_chirp_analytic_mask(e.a, F, e.P, (float)inv_p, e.st), "analytic_mask") &&
      e.convolve() &&
      e.ok(launch_chirp_store(e.a, e.out, n, F, (float)(inv_p * inv_f), 1, e.sum, inv_f, e.st),
           "chirp_store") &&
      e.ok(hipMemcpyAsync(out, e.out, sizeof(float2) * n, hipMemcpyDeviceToHost, e.st), "copy out") &&
      e.ok(hipStreamSynchronize(e.st), "sync");
  return good ? GCWT_OK : engine_error(e);
}

extern "C" {

int gcwt_dft(const float* x, int64_t n, int is_complex, int inverse, float* out, int device) {
  return guarded([&] { return dft_impl(x, n, is_complex, inverse, out, device); });
}

int gcwt_analytic_signal(const float* signal, int64_t n, int64_t fft_length, float* out,
                         int device) {
  return guarded([&] { return analytic_impl(signal, n, fft_length, out, device); });
}

// ---- the same operators in float64 (ops64.hip): the reference's own arithmetic and result dtype ----
int gcwt_dft_f64(const double* x, int64_t n, int is_complex, int inverse, double* out, int device) {
  return guarded([&] {
    if (!x || !out || n < 1) return fail(GCWT_ERR_INVALID, "bad arguments");
    if (n > ((int64_t)1 << 23)) return fail(GCWT_ERR_UNSUPPORTED, "float64 DFT lengths up to 2^23");
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return fail(GCWT_ERR_NO_DEVICE, "cannot select the device");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) { (void)hipGetLastError(); return fail(GCWT_ERR_NO_DEVICE, "no HIP device"); }
    const hipError_t e = gcwt::dft_f64(x, n, is_complex, inverse, out);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? GCWT_ERR_NOMEM : GCWT_ERR_HIP, std::string("float64 DFT: ") + hipGetErrorString(e));
    return (int)GCWT_OK;
  });
}

int gcwt_fastconv_f64(const double* signal, int64_t n, int signal_is_complex, const double* kernel, int64_t m,
                      int kernel_is_complex, int mode, double* out, int device) {
  return guarded([&] {
    if (!signal || !kernel || !out || n < 1 || m < 1 || mode < 0 || mode > 2) return fail(GCWT_ERR_INVALID, "bad arguments");
    if (mode == 2 && n < m) return fail(GCWT_ERR_INVALID, "'valid' needs a signal at least as long as the kernel");
    // (results beyond one 2^24-point transform are made by overlap-add over chunks of the signal, as convolution.py:70-77
    // does; the kernel's own transform bounds its length)
    if (n + m - 1 > ((int64_t)1 << 24) && m > ((int64_t)1 << 23)) return fail(GCWT_ERR_UNSUPPORTED, "float64 convolutions with kernels up to 2^23 taps");
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return fail(GCWT_ERR_NO_DEVICE, "cannot select the device");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) { (void)hipGetLastError(); return fail(GCWT_ERR_NO_DEVICE, "no HIP device"); }
    // 'same' is centred as convolution.py:85 does it: (total - n) // 2 samples dropped in front
    const int64_t first = mode == 0 ? 0 : (mode == 1 ? (m - 1) / 2 : m - 1);
    const int64_t cnt = mode == 0 ? n + m - 1 : (mode == 1 ? n : n - m + 1);
    const hipError_t e = gcwt::fastconv_f64(signal, n, signal_is_complex, kernel, m, kernel_is_complex, first, cnt, out);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? GCWT_ERR_NOMEM : GCWT_ERR_HIP, std::string("float64 convolution: ") + hipGetErrorString(e));
    return (int)GCWT_OK;
  });
}

}  // extern "C"
