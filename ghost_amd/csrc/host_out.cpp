#include "host_out.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace gcwt {

namespace {

// [begin, end) split over up to 8 threads; small jobs stay on the caller's thread
template <typename F>
void parallel_rows(size_t n_rows, size_t bytes, F&& body) {
  unsigned hw = std::thread::hardware_concurrency();
  size_t n_thr = std::min<size_t>({size_t(hw ? hw : 1), size_t(8), n_rows});
  if (bytes < (size_t(1) << 20)) n_thr = 1;
  if (n_thr <= 1) { body(size_t(0), n_rows); return; }
  std::vector<std::thread> pool;
  pool.reserve(n_thr - 1);
  const size_t per = (n_rows + n_thr - 1) / n_thr;
  for (size_t t = 1; t < n_thr; ++t) {
    const size_t a = std::min(n_rows, t * per), b = std::min(n_rows, a + per);
    if (a < b) pool.emplace_back([&body, a, b] { body(a, b); });
  }
  body(size_t(0), std::min(n_rows, per));
  for (auto& th : pool) th.join();
}

void widen_row(const float* src, double* dst, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}

}  // namespace

HostOut::~HostOut() { release(); }

void HostOut::release() {
  for (int i = 0; i < 2; ++i) {
    if (ring_[i]) { (void)hipHostFree(ring_[i]); ring_[i] = nullptr; }
    if (landed_[i]) { (void)hipEventDestroy(landed_[i]); landed_[i] = nullptr; }
  }
}

hipError_t HostOut::drain(const float* d_src, size_t src_pitch_floats, size_t n_rows,
                          size_t row_floats, void* dst, bool widen, hipStream_t stream) {
  if (n_rows == 0 || row_floats == 0) return hipStreamSynchronize(stream);
  hipError_t e;
  for (int i = 0; i < 2; ++i) {
    if (!ring_[i] && (e = hipHostMalloc((void**)&ring_[i], kChunkBytes, hipHostMallocDefault)) != hipSuccess)
      return e;
    if (!landed_[i] && (e = hipEventCreateWithFlags(&landed_[i], hipEventDisableTiming)) != hipSuccess)
      return e;
  }
  const size_t row_bytes = row_floats * sizeof(float);
  if (row_bytes > kChunkBytes) {
    // rows longer than a staging buffer (> 8 M samples): plain strided copy, widen in place
    // is not possible -- the caller falls back to float32 + its own conversion
    return hipErrorInvalidValue;
  }
  const size_t rows_per = std::max<size_t>(1, kChunkBytes / row_bytes);
  const size_t n_chunks = (n_rows + rows_per - 1) / rows_per;
  auto issue = [&](size_t c) -> hipError_t {
    const size_t r0 = c * rows_per, nr = std::min(rows_per, n_rows - r0);
    hipError_t er = hipMemcpy2DAsync(ring_[c & 1], row_bytes, d_src + r0 * src_pitch_floats,
                                     src_pitch_floats * sizeof(float), row_bytes, nr,
                                     hipMemcpyDeviceToHost, stream);
    if (er != hipSuccess) return er;
    return hipEventRecord(landed_[c & 1], stream);
  };
  if ((e = issue(0)) != hipSuccess) return e;
  for (size_t c = 0; c < n_chunks; ++c) {
    if (c + 1 < n_chunks && (e = issue(c + 1)) != hipSuccess) return e;   // other buffer: free
    if ((e = hipEventSynchronize(landed_[c & 1])) != hipSuccess) return e;
    const size_t r0 = c * rows_per, nr = std::min(rows_per, n_rows - r0);
    const float* src = ring_[c & 1];
    if (widen) {
      double* out = static_cast<double*>(dst) + r0 * row_floats;
      parallel_rows(nr, nr * row_bytes, [=](size_t a, size_t b) {
        widen_row(src + a * row_floats, out + a * row_floats, (b - a) * row_floats);
      });
    } else {
      float* out = static_cast<float*>(dst) + r0 * row_floats;
      parallel_rows(nr, nr * row_bytes, [=](size_t a, size_t b) {
        std::memcpy(out + a * row_floats, src + a * row_floats, (b - a) * row_bytes);
      });
    }
  }
  return hipStreamSynchronize(stream);
}

}  // namespace gcwt
