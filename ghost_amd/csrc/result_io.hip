// result_io.hip -- a device-resident result on its way to the host (include/ghostcwt.h: gcwt_host_alloc,
// gcwt_rows_to_host).  The reference hands back whole float64 arrays (transforms.py:203-204, 496-527); here the
// result stays on the device after transform() and is brought over when -- and as far as -- it is asked for:
//   * into page-locked memory (gcwt_host_alloc) the DMA engines write at the link's rate, no staging copy and no
//     page faults (a fresh 856 MB NumPy array costs more in first-touch faults than the 428 MB cost on the wire);
//   * float64 results cross the link as float32 -- half the bytes -- and a standing pool of host threads widens each
//     tile into the destination with streaming stores while the next tiles are on the wire (host_out.cpp); option
//     host_widen = 0 widens on the device instead and sends float64 (17 against 10.5 ms for 800 MB on a 55 GB/s link);
//   * any rectangle of (row, sample) goes by itself: rows `src_pitch` apart on the device, `dst_pitch` on the host.
// Destinations that are not page-locked go through the plan-independent staging ring (host_out.cpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>

#include "host_out.h"
#include "kernels.h"
#include "options.h"

namespace gcwt {

namespace {

__global__ void __launch_bounds__(256) k_widen_rows(const float* __restrict__ src, int64_t src_pitch,
                                                    double* __restrict__ dst, int64_t row_elems, int64_t n) {
  // n = rows * row_elems elements of the chunk, dense in dst; 4 per thread
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = i0 + k;
    if (i < n) {
      const int64_t r = i / row_elems, c = i - r * row_elems;
      dst[i] = (double)src[r * src_pitch + c];
    }
  }
}

// one per device: a process that drives several (ghost_amd/multi.py: one host thread per device slot) drains them side
// by side, each over its own link; two drains from the same device take turns
struct ResultIo {
  std::mutex mu;
  hipStream_t stream = nullptr;
  double* d_stage = nullptr;            // widened chunk on the device
  size_t stage_elems = 0;
  HostOut ring;                         // destinations that are not page-locked
};
constexpr int kMaxDevices = 64;
ResultIo& io(int device) {
  static ResultIo* const s = new ResultIo[kMaxDevices];      // never destroyed: nothing of HIP runs at process exit
  return s[device];
}
constexpr size_t kStageElems = (size_t)16 << 20;   // 128 MB of doubles

}  // namespace

hipError_t rows_to_host(const float* d_src, int64_t src_pitch, int64_t n_rows, int64_t row_elems, void* dst,
                        int64_t dst_pitch, bool widen, bool pinned) {
  if (n_rows <= 0 || row_elems <= 0) return hipSuccess;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, d_src);
  if (e != hipSuccess) return e;
  if (attr.device < 0 || attr.device >= kMaxDevices) return hipErrorInvalidDevice;
  if ((e = hipSetDevice(attr.device)) != hipSuccess) return e;      // (the calling thread's device, from here on)
  ResultIo& s = io(attr.device);
  std::lock_guard<std::mutex> lock(s.mu);
  if (!s.stream && (e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess) return e;
  if (!pinned) {
    if (dst_pitch != row_elems) return hipErrorInvalidValue;   // the staging ring scatters into dense rows
    return s.ring.drain(d_src, (size_t)src_pitch, (size_t)n_rows, (size_t)row_elems, dst, widen, s.stream);
  }
  if (!widen) {
    if (src_pitch == row_elems && dst_pitch == row_elems)
      e = hipMemcpyAsync(dst, d_src, sizeof(float) * (size_t)n_rows * (size_t)row_elems, hipMemcpyDeviceToHost, s.stream);
    else
      e = hipMemcpy2DAsync(dst, sizeof(float) * (size_t)dst_pitch, d_src, sizeof(float) * (size_t)src_pitch,
                           sizeof(float) * (size_t)row_elems, (size_t)n_rows, hipMemcpyDeviceToHost, s.stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s.stream);
  }
  if (option_or("host_widen", 1) != 0)     // float32 over the link, widened by host threads (0: on the device, below)
    return s.ring.drain(d_src, (size_t)src_pitch, (size_t)n_rows, (size_t)row_elems, dst, true, s.stream, (size_t)dst_pitch);
  // float64: chunks of whole rows (or pieces of one long row) widened into the staging buffer, then over the link;
  // the stream keeps widen(c + 1) behind copy(c), and a widening pass is a hundredth of its copy
  if (!s.d_stage) {
    if ((e = hipMalloc((void**)&s.d_stage, sizeof(double) * kStageElems)) != hipSuccess) return e;
    s.stage_elems = kStageElems;
  }
  const int64_t cols_per = std::min<int64_t>(row_elems, (int64_t)s.stage_elems);
  const int64_t rows_per = std::max<int64_t>(1, (int64_t)s.stage_elems / cols_per);
  for (int64_t r0 = 0; r0 < n_rows; r0 += rows_per) {
    const int64_t nr = std::min(rows_per, n_rows - r0);
    for (int64_t c0 = 0; c0 < row_elems; c0 += cols_per) {
      const int64_t nc = std::min(cols_per, row_elems - c0);
      const int64_t n = nr * nc;
      hipLaunchKernelGGL(k_widen_rows, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s.stream,
                         d_src + r0 * src_pitch + c0, src_pitch, s.d_stage, nc, n);
      if ((e = hipGetLastError()) != hipSuccess) return e;
      double* out = static_cast<double*>(dst) + r0 * dst_pitch + c0;
      if (nc == dst_pitch)
        e = hipMemcpyAsync(out, s.d_stage, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s.stream);
      else
        e = hipMemcpy2DAsync(out, sizeof(double) * (size_t)dst_pitch, s.d_stage, sizeof(double) * (size_t)nc,
                             sizeof(double) * (size_t)nc, (size_t)nr, hipMemcpyDeviceToHost, s.stream);
      if (e != hipSuccess) return e;
    }
  }
  return hipStreamSynchronize(s.stream);
}

}  // namespace gcwt
