// conv_plan.cpp -- the operator layer under the transform as a reusable GPU operator:
// FFT convolution of a batch of real signals with one kernel given in the time domain
// (ghost/sigtools/convolution.py:16-216, fastconv_scipy / fastconv_fftw) or by its DFT
// (:218-402, fastconv_freq_scipy / fastconv_freq_fftw).
//
// A plan owns its stream, twiddle tables, kernel spectrum and workspace; signals of any
// length run as overlap-save chunks of one power-of-two FFT (the reference's chunked
// overlap-add, convolution.py:68-77, with the history read instead of added): chunk c
// holds input samples [c*step - (m-1), c*step + step), step = P - (m-1), and yields
// full-convolution samples [c*step, c*step + step).  Chunks x channels are batched through
// the same two-pass FFT kernels the transform uses, up to 16 chunks per launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "../../include/ghostcwt.h"
#include "kernels.h"
#include "planner.h"

using namespace gcwt;

int gcwt_internal_set_error(int code, const char* msg);   // api.cpp (C++ linkage)

struct gcwt_conv_plan {
  int64_t n = 0, m = 0, p = 0, step = 0, n_chunks = 0;
  int32_t n_channels = 1, p1 = 0, device = -1;
  bool have_kernel = false;
  hipStream_t st = nullptr;
  float2 *tw4096 = nullptr, *tw256 = nullptr, *hspec = nullptr, *work = nullptr, *ktmp = nullptr;
  double* zero = nullptr;        // [C] zeros: the FFT kernels subtract sums[ch] * inv_n
  float* d_in = nullptr;
  float2* d_out = nullptr;
  size_t in_bytes = 0, out_bytes = 0;
  int chunks_per_batch = 1;
};

namespace {

int fail(int code, const std::string& msg) { return gcwt_internal_set_error(code, msg.c_str()); }
int hip_fail(hipError_t e, const char* what) {
  return fail(e == hipErrorOutOfMemory ? GCWT_ERR_NOMEM : GCWT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define CP_TRY(call)                                     \
  do {                                                   \
    hipError_t e_ = (call);                              \
    if (e_ != hipSuccess) return hip_fail(e_, #call);    \
  } while (0)

void release(gcwt_conv_plan* p) {
  (void)hipFree(p->tw4096); (void)hipFree(p->tw256); (void)hipFree(p->hspec); (void)hipFree(p->work);
  (void)hipFree(p->ktmp); (void)hipFree(p->zero); (void)hipFree(p->d_in); (void)hipFree(p->d_out);
  if (p->st) (void)hipStreamDestroy(p->st);
}

// forward FFT of `slots` P-point arrays already in `v` as complex (k1-major result, all bins)
int forward_complex(gcwt_conv_plan* p, float2* v, int slots) {
  CP_TRY(launch_fft_cols(-1, false, v, v, p->p1, kRowLen, p->p, p->p, p->p1 > 1 ? p->p : 0, p->tw4096,
                         p->tw256, p->zero, 0.0, 0, slots, p->st));
  CP_TRY(launch_fft_rows(-1, v, v, kRowLen, p->p1, kRowLen, kRowLen, p->p, p->p, 0, p->tw4096, p->tw256,
                         1.0f, slots, p->st));
  return GCWT_OK;
}

}  // namespace

namespace gcwt {
hipError_t launch_conv_permute(const float2* natural, float2* k1major, int p1, hipStream_t st);
hipError_t launch_conv_widen(const float* src, float2* dst, int64_t m, hipStream_t st);
hipError_t launch_conv_store(const float2* y, float2* out, int64_t p, int64_t m, int64_t step,
                             int chunk0, int n_chunks_here, int n_channels, int64_t first,
                             int64_t count, float scale, hipStream_t st);
}  // namespace gcwt

extern "C" {

int gcwt_conv_plan_create(gcwt_conv_plan** out, int64_t n, int64_t m, int32_t n_channels,
                          int32_t fft_log2, int32_t device) {
  if (!out) return fail(GCWT_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (n <= 0 || m <= 0 || n_channels <= 0) return fail(GCWT_ERR_INVALID, "signal, kernel and channel counts must be positive");
  if (n_channels > 4095) return fail(GCWT_ERR_UNSUPPORTED, "more than 4095 channels per convolution plan");
  if (fft_log2 != 0 && (fft_log2 < 12 || fft_log2 > 22)) return fail(GCWT_ERR_INVALID, "fft_log2 must be 0 or 12..22");
  int64_t P = kRowLen;
  if (fft_log2 == 0) {
    // One FFT when the whole convolution fits: chunk 0 of the overlap-save layout starts with
    // m - 1 zeros of history, so a single chunk needs n + 2 (m - 1) points (with only n + m - 1 the
    // work would be two full-size chunks).  Else overlap-save chunks of 2^22.
    while (P < n + 2 * (m - 1) && P < ((int64_t)1 << 22)) P <<= 1;
  } else {
    P = (int64_t)1 << fft_log2;
  }
  if (P < m) return fail(GCWT_ERR_INVALID, "FFT length must be at least the kernel size");   // convolution.py:61
  if (P - (m - 1) < P / 8 && n + m - 1 > P)
    return fail(GCWT_ERR_UNSUPPORTED, "kernel too long for overlap-save chunks of 2^22 points");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(GCWT_ERR_NO_DEVICE, "no HIP device: libghostcwt has no CPU path");
  if (device >= 0) CP_TRY(hipSetDevice(device));
  gcwt_conv_plan* p = new (std::nothrow) gcwt_conv_plan();
  if (!p) return fail(GCWT_ERR_NOMEM, "out of host memory");
  try {
  p->n = n; p->m = m; p->p = P; p->p1 = (int)(P / kRowLen); p->n_channels = n_channels; p->device = device;
  p->step = P - (m - 1);
  p->n_chunks = (n + m - 1 + p->step - 1) / p->step;
  // workspace: as many chunks at a time as 2 GiB hold, at most 16 (kernels.h: kSegBatch)
  const int64_t per_chunk = (int64_t)sizeof(float2) * P * n_channels;
  p->chunks_per_batch = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)kSegBatch, p->n_chunks,
                                                                      ((int64_t)2 << 30) / per_chunk,
                                                                      (int64_t)65535 / n_channels}));
  auto bail = [&](int rc) { release(p); delete p; return rc; };
#define CP_B(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bail(hip_fail(e_, #call)); } while (0)
  CP_B(hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking));
  CP_B(hipMalloc((void**)&p->tw4096, sizeof(float2) * (kRowLen / 2)));
  CP_B(hipMalloc((void**)&p->tw256, sizeof(float2) * 256));
  CP_B(hipMalloc((void**)&p->hspec, sizeof(float2) * P));
  CP_B(hipMalloc((void**)&p->ktmp, sizeof(float2) * P));
  CP_B(hipMalloc((void**)&p->work, (size_t)per_chunk * p->chunks_per_batch));
  CP_B(hipMalloc((void**)&p->zero, sizeof(double) * n_channels));
  std::vector<float2> t4(kRowLen / 2), t2(256);
  for (int j = 0; j < kRowLen / 2; ++j) {
    const double a = -2.0 * M_PI * j / kRowLen;
    t4[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  for (int q = 0; q < 256; ++q) {
    const double a = 2.0 * M_PI * q / 256.0;
    t2[q] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  CP_B(hipMemcpyAsync(p->tw4096, t4.data(), sizeof(float2) * t4.size(), hipMemcpyHostToDevice, p->st));
  CP_B(hipMemcpyAsync(p->tw256, t2.data(), sizeof(float2) * 256, hipMemcpyHostToDevice, p->st));
  CP_B(hipMemsetAsync(p->zero, 0, sizeof(double) * n_channels, p->st));
  CP_B(hipStreamSynchronize(p->st));
#undef CP_B
  } catch (const std::exception&) {      // nothing unwinds across the C ABI
    release(p);
    delete p;
    return fail(GCWT_ERR_NOMEM, "out of host memory");
  }
  *out = p;
  return GCWT_OK;
}

void gcwt_conv_plan_destroy(gcwt_conv_plan* p) {
  if (!p) return;
  if (p->device >= 0) (void)hipSetDevice(p->device);
  release(p);
  delete p;
}

int gcwt_conv_plan_info(const gcwt_conv_plan* p, int64_t* fft_length, int64_t* chunk, int64_t* n_chunks) {
  if (!p) return fail(GCWT_ERR_INVALID, "NULL plan");
  if (fft_length) *fft_length = p->p;
  if (chunk) *chunk = p->step;
  if (n_chunks) *n_chunks = p->n_chunks;
  return GCWT_OK;
}

int gcwt_conv_plan_set_kernel(gcwt_conv_plan* p, const float* kernel, int is_complex, int on_device) {
  if (!p || !kernel) return fail(GCWT_ERR_INVALID, "NULL argument");
  if (p->device >= 0) CP_TRY(hipSetDevice(p->device));
  CP_TRY(hipMemsetAsync(p->hspec, 0, sizeof(float2) * p->p, p->st));
  if (is_complex) {
    CP_TRY(hipMemcpyAsync(p->hspec, kernel, sizeof(float2) * p->m, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, p->st));
  } else {
    // real taps: onto the device as they are (scratch array), widened to complex there
    CP_TRY(hipMemcpyAsync(p->ktmp, kernel, sizeof(float) * p->m, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, p->st));
    CP_TRY(launch_conv_widen(reinterpret_cast<const float*>(p->ktmp), p->hspec, p->m, p->st));
  }
  int rc = forward_complex(p, p->hspec, 1);
  if (rc) return rc;
  CP_TRY(hipStreamSynchronize(p->st));    // the caller's host kernel array may go away
  p->have_kernel = true;
  return GCWT_OK;
}

int gcwt_conv_plan_set_kernel_fd(gcwt_conv_plan* p, const float* kernel_fd, int on_device) {
  if (!p || !kernel_fd) return fail(GCWT_ERR_INVALID, "NULL argument");
  if (p->device >= 0) CP_TRY(hipSetDevice(p->device));
  CP_TRY(hipMemcpyAsync(p->ktmp, kernel_fd, sizeof(float2) * p->p, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, p->st));
  CP_TRY(launch_conv_permute(p->ktmp, p->hspec, p->p1, p->st));
  CP_TRY(hipStreamSynchronize(p->st));
  p->have_kernel = true;
  return GCWT_OK;
}

int gcwt_conv_plan_execute(gcwt_conv_plan* p, const float* signal, int mode, float* out, int flags) {
  if (!p || !signal || !out) return fail(GCWT_ERR_INVALID, "NULL argument");
  if (!p->have_kernel) return fail(GCWT_ERR_INVALID, "no kernel set on this convolution plan");
  if (mode < 0 || mode > 2) return fail(GCWT_ERR_INVALID, "Mode must be 'full', 'same', or 'valid'");
  if (mode == 2 && p->n < p->m)
    return fail(GCWT_ERR_INVALID, "Cannot do a 'valid' convolution because the input is shorter than the kernel");
  if (p->device >= 0) CP_TRY(hipSetDevice(p->device));
  const int64_t n = p->n, m = p->m, P = p->p, total = n + m - 1;
  const int C = p->n_channels;
  const int64_t count = mode == 0 ? total : (mode == 1 ? n : n - m + 1);
  const int64_t first = (total - count) / 2;                      // convolution.py:79-87
  const float* dx = signal;
  float2* dout = reinterpret_cast<float2*>(out);
  if (!(flags & GCWT_X_ON_DEVICE)) {
    const size_t need = sizeof(float) * (size_t)C * (size_t)n;
    if (p->in_bytes < need) {
      (void)hipFree(p->d_in); p->d_in = nullptr; p->in_bytes = 0;
      CP_TRY(hipMalloc((void**)&p->d_in, need));
      p->in_bytes = need;
    }
    CP_TRY(hipMemcpyAsync(p->d_in, signal, need, hipMemcpyHostToDevice, p->st));
    dx = p->d_in;
  }
  if (!(flags & GCWT_OUT_ON_DEVICE)) {
    const size_t need = sizeof(float2) * (size_t)C * (size_t)count;
    if (p->out_bytes < need) {
      (void)hipFree(p->d_out); p->d_out = nullptr; p->out_bytes = 0;
      CP_TRY(hipMalloc((void**)&p->d_out, need));
      p->out_bytes = need;
    }
    dout = p->d_out;
  }
  for (int64_t c0 = 0; c0 < p->n_chunks; c0 += p->chunks_per_batch) {
    const int nb = (int)std::min<int64_t>(p->chunks_per_batch, p->n_chunks - c0);
    SegIn sin{};
    sin.n_channels = C;
    for (int g = 0; g < nb; ++g) {
      const int64_t s0 = (c0 + g) * p->step - (m - 1);            // signal index of segment sample 0
      sin.x_off[g] = s0;
      sin.n_lead[g] = std::max<int64_t>(0, -s0);
      sin.n_valid[g] = std::min<int64_t>(P, n - s0);
      // (a chunk exists only while c*step < n + m - 1, i.e. s0 < n: at least one real sample)
    }
    const int slots = C * nb;
    CP_TRY(launch_fft_cols_batch(dx, p->work, p->p1, kRowLen, n, P, p->p1 > 1 ? P : 0, p->tw4096, p->tw256,
                                 p->zero, 0.0, sin, nb, p->st, p->p1));
    CP_TRY(launch_fft_rows(-1, p->work, p->work, kRowLen, p->p1, kRowLen, kRowLen, P, P, 0, p->tw4096,
                           p->tw256, 1.0f, slots, p->st));
    CP_TRY(launch_fullband_mul(p->work, p->hspec, p->work, P, slots, p->st));
    CP_TRY(launch_fft_rows(+1, p->work, p->work, kRowLen, p->p1, kRowLen, kRowLen, P, P, p->p1 > 1 ? P : 0,
                           p->tw4096, p->tw256, 1.0f, slots, p->st));
    if (p->p1 > 1)
      CP_TRY(launch_fft_cols(+1, false, p->work, p->work, p->p1, kRowLen, P, P, 0, p->tw4096, p->tw256,
                             p->zero, 0.0, 0, slots, p->st));
    CP_TRY(launch_conv_store(p->work, dout, P, m, p->step, (int)c0, nb, C, first, count,
                             (float)(1.0 / (double)P), p->st));
  }
  if (!(flags & GCWT_OUT_ON_DEVICE))
    CP_TRY(hipMemcpyAsync(out, dout, sizeof(float2) * (size_t)C * (size_t)count, hipMemcpyDeviceToHost, p->st));
  CP_TRY(hipStreamSynchronize(p->st));
  return GCWT_OK;
}

// One-shot form (the round-1 entry point): a plan for this call only.
int gcwt_fastconv(const float* signal, int64_t n, const float* kernel, int64_t m, int kernel_is_complex,
                  int mode, float* out, int device) {
  if (!signal || !kernel || !out || n <= 0 || m <= 0) return fail(GCWT_ERR_INVALID, "bad argument");
  if (mode < 0 || mode > 2) return fail(GCWT_ERR_INVALID, "Mode must be 'full', 'same', or 'valid'");
  if (mode == 2 && n < m)
    return fail(GCWT_ERR_INVALID, "Cannot do a 'valid' convolution because the input is shorter than the kernel");
  gcwt_conv_plan* p = nullptr;
  int rc = gcwt_conv_plan_create(&p, n, m, 1, 0, device);
  if (rc) return rc;
  rc = gcwt_conv_plan_set_kernel(p, kernel, kernel_is_complex, 0);
  if (!rc) rc = gcwt_conv_plan_execute(p, signal, mode, out, 0);
  gcwt_conv_plan_destroy(p);
  return rc;
}

}  // extern "C"
