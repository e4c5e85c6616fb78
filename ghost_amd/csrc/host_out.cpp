#include "host_out.h"
#include "options.h"

#include <algorithm>
#include <cstdlib>
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
  // Tiles of the result: whole rows while they fit a staging buffer, else pieces of one
  // row (recordings longer than 8 M samples), so any row length streams through the ring.
  size_t chunk_floats = kChunkBytes / sizeof(float);
  if (option_is_set("stage_floats"))                        // tests: force small tiles
    chunk_floats = std::min(chunk_floats, std::max<size_t>(64, (size_t)option_or("stage_floats", 0)));
  const size_t cols_per = std::min(row_floats, chunk_floats);
  const size_t rows_per = std::max<size_t>(1, chunk_floats / cols_per);
  const size_t col_tiles = (row_floats + cols_per - 1) / cols_per;
  const size_t row_tiles = (n_rows + rows_per - 1) / rows_per;
  const size_t n_chunks = row_tiles * col_tiles;
  struct Tile { size_t r0, nr, c0, nc; };
  auto tile = [&](size_t c) {
    const size_t rt = c / col_tiles, ct = c % col_tiles;
    Tile t{rt * rows_per, 0, ct * cols_per, 0};
    t.nr = std::min(rows_per, n_rows - t.r0);
    t.nc = std::min(cols_per, row_floats - t.c0);
    return t;
  };
  auto issue = [&](size_t c) -> hipError_t {
    const Tile t = tile(c);
    hipError_t er = hipMemcpy2DAsync(ring_[c & 1], t.nc * sizeof(float),
                                     d_src + t.r0 * src_pitch_floats + t.c0,
                                     src_pitch_floats * sizeof(float), t.nc * sizeof(float), t.nr,
                                     hipMemcpyDeviceToHost, stream);
    if (er != hipSuccess) return er;
    return hipEventRecord(landed_[c & 1], stream);
  };
  if ((e = issue(0)) != hipSuccess) return e;
  for (size_t c = 0; c < n_chunks; ++c) {
    if (c + 1 < n_chunks && (e = issue(c + 1)) != hipSuccess) return e;   // other buffer: free
    if ((e = hipEventSynchronize(landed_[c & 1])) != hipSuccess) return e;
    const Tile t = tile(c);
    const float* src = ring_[c & 1];
    if (widen) {
      double* out = static_cast<double*>(dst) + t.r0 * row_floats + t.c0;
      parallel_rows(t.nr, t.nr * t.nc * sizeof(float), [=](size_t a, size_t b) {
        for (size_t r = a; r < b; ++r) widen_row(src + r * t.nc, out + r * row_floats, t.nc);
      });
    } else {
      float* out = static_cast<float*>(dst) + t.r0 * row_floats + t.c0;
      parallel_rows(t.nr, t.nr * t.nc * sizeof(float), [=](size_t a, size_t b) {
        for (size_t r = a; r < b; ++r)
          std::memcpy(out + r * row_floats, src + r * t.nc, t.nc * sizeof(float));
      });
    }
  }
  return hipStreamSynchronize(stream);
}

}  // namespace gcwt
