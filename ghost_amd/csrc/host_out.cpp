#include "host_out.h"
#include "options.h"

#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cctype>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace gcwt {

namespace {

// Workers that outlive a call: a result of 400 MB goes by in two dozen tiles, and sixteen threads made and joined
// for each tile cost more than the tile's widening.  One job at a time (callers hold the drain for its length).
class Workers {
 public:
  static Workers& get() {
    // never destroyed (no join while the process exits); a forked child starts its own (the parent's threads are not there)
    static std::once_flag fork_hook;
    std::call_once(fork_hook, [] { pthread_atfork(nullptr, nullptr, [] { instance().store(nullptr); }); });
    Workers* w = instance().load();
    if (!w) {
      Workers* fresh = new Workers();
      if (instance().compare_exchange_strong(w, fresh)) w = fresh;
      else delete fresh;
    }
    return *w;
  }
  // The workers run beside the device: the staging buffers and page-locked destinations live on the device's NUMA node
  // (the runtime places them there), and threads on the other socket widen at half the rate (measured: 800 MB in
  // 7 against 17 ms).  Best effort: within the process's own affinity, nothing if the node cannot be told.
  void bind(const cpu_set_t& cpus) {
    std::lock_guard<std::mutex> one(job_mu_);
    cpus_ = cpus;
    bound_ = true;
    for (std::thread& t : threads_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(cpus_), &cpus_);
  }
  // body(a, b) over [0, n) in contiguous pieces, the caller's thread taking one of them
  void run(size_t n, size_t n_thr, const std::function<void(size_t, size_t)>& body) {
    n_thr = std::min(n_thr, n);
    if (n_thr <= 1) { body(0, n); return; }
    std::lock_guard<std::mutex> one(job_mu_);
    grow(n_thr - 1);
    const size_t per = (n + n_thr - 1) / n_thr;
    {
      std::lock_guard<std::mutex> lock(mu_);
      body_ = &body;
      n_ = n; per_ = per;
      next_ = 1; last_ = n_thr;
      pending_ = n_thr - 1;
      ++epoch_;
    }
    cv_.notify_all();
    body(0, std::min(n, per));
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [this] { return pending_ == 0; });
    body_ = nullptr;
  }

 private:
  static std::atomic<Workers*>& instance() {
    static std::atomic<Workers*> p{nullptr};
    return p;
  }
  void grow(size_t n) {
    while (threads_.size() < n) {
      threads_.emplace_back([this] { loop(); });
      if (bound_) (void)pthread_setaffinity_np(threads_.back().native_handle(), sizeof(cpus_), &cpus_);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      size_t piece;
      const std::function<void(size_t, size_t)>* body;
      size_t n, per;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return epoch_ != seen && next_ < last_; });
        piece = next_++;
        if (next_ >= last_) seen = epoch_;    // the job's pieces are all taken: wait for the next job
        body = body_; n = n_; per = per_;
      }
      const size_t a = std::min(n, piece * per), b = std::min(n, a + per);
      if (a < b) (*body)(a, b);
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;        // never joined: the object is never destroyed
  cpu_set_t cpus_;
  bool bound_ = false;
  const std::function<void(size_t, size_t)>* body_ = nullptr;
  size_t n_ = 0, per_ = 0, next_ = 0, last_ = 0, pending_ = 0;
  uint64_t epoch_ = 0;
};

// CPUs on the current device's NUMA node (sysfs: local_cpulist of its PCI function), cut to the process's affinity
void bind_workers_beside_device() {
  static std::atomic<int> bound_device{-2};       // (drains of different plans may run on different threads)
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev == bound_device.load()) return;
  bound_device.store(dev);
  char bdf[64] = {0}, path[160];
  if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), dev) != hipSuccess) { (void)hipGetLastError(); return; }
  for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return;
  char line[4096] = {0};
  const bool ok = fgets(line, sizeof(line), f) != nullptr;
  fclose(f);
  if (!ok) return;
  cpu_set_t node, mine, both;
  CPU_ZERO(&node);
  for (char* tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
    int a = 0, b = 0;
    const int n = sscanf(tok, "%d-%d", &a, &b);
    if (n < 1) continue;
    if (n == 1) b = a;
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET(c, &node);
  }
  if (sched_getaffinity(0, sizeof(mine), &mine) != 0) return;
  CPU_AND(&both, &node, &mine);
  if (CPU_COUNT(&both) == 0 || CPU_EQUAL(&both, &mine)) return;     // nothing to choose
  Workers::get().bind(both);
}

size_t worker_count(size_t bytes) {
  if (bytes < (size_t(1) << 20)) return 1;
  const unsigned hw = std::thread::hardware_concurrency();
  return std::min<size_t>(size_t(hw ? hw : 1), (size_t)std::max<long long>(1, option_or("host_threads", 8)));
}

// float32 -> float64 with streaming stores where the CPU has them: the destination is written once and not read by
// these threads, and an ordinary store first reads the line it is about to overwrite -- half as much memory traffic.
__attribute__((target("avx2"))) void widen_row_avx2(const float* src, double* dst, size_t n) {
  size_t i = 0;
  while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31)) { dst[i] = (double)src[i]; ++i; }
  for (; i + 8 <= n; i += 8) {
    const __m256 v = _mm256_loadu_ps(src + i);
    _mm256_stream_pd(dst + i, _mm256_cvtps_pd(_mm256_castps256_ps128(v)));
    _mm256_stream_pd(dst + i + 4, _mm256_cvtps_pd(_mm256_extractf128_ps(v, 1)));
  }
  for (; i < n; ++i) dst[i] = (double)src[i];
  _mm_sfence();
}
void widen_row_plain(const float* src, double* dst, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}
void widen_row(const float* src, double* dst, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) widen_row_avx2(src, dst, n);
  else widen_row_plain(src, dst, n);
}

}  // namespace

HostOut::~HostOut() { release(); }

void HostOut::release() {
  for (int i = 0; i < kDepth; ++i) {
    if (ring_[i]) { (void)hipHostFree(ring_[i]); ring_[i] = nullptr; }
    if (landed_[i]) { (void)hipEventDestroy(landed_[i]); landed_[i] = nullptr; }
  }
}

hipError_t HostOut::drain(const float* d_src, size_t src_pitch_floats, size_t n_rows,
                          size_t row_floats, void* dst, bool widen, hipStream_t stream, size_t dst_pitch) {
  if (n_rows == 0 || row_floats == 0) return hipStreamSynchronize(stream);
  if (dst_pitch == 0) dst_pitch = row_floats;
  bind_workers_beside_device();
  hipError_t e;
  for (int i = 0; i < kDepth; ++i) {
    if (!ring_[i] && (e = hipHostMalloc((void**)&ring_[i], kChunkBytes, hipHostMallocDefault)) != hipSuccess)
      return e;
    if (!landed_[i] && (e = hipEventCreateWithFlags(&landed_[i], hipEventDisableTiming)) != hipSuccess)
      return e;
  }
  // Tiles of the result: whole rows while they fit a staging buffer, else pieces of one
  // row (recordings longer than 4 M samples), so any row length streams through the ring.
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
    hipError_t er = hipMemcpy2DAsync(ring_[c % kDepth], t.nc * sizeof(float),
                                     d_src + t.r0 * src_pitch_floats + t.c0,
                                     src_pitch_floats * sizeof(float), t.nc * sizeof(float), t.nr,
                                     hipMemcpyDeviceToHost, stream);
    if (er != hipSuccess) return er;
    return hipEventRecord(landed_[c % kDepth], stream);
  };
  // kDepth - 1 tiles on the wire while the workers are on one
  for (size_t c = 0; c < std::min<size_t>(n_chunks, kDepth - 1); ++c)
    if ((e = issue(c)) != hipSuccess) return e;
  for (size_t c = 0; c < n_chunks; ++c) {
    if (c + kDepth - 1 < n_chunks && (e = issue(c + kDepth - 1)) != hipSuccess) return e;   // the buffer tile c - 1 left
    if ((e = hipEventSynchronize(landed_[c % kDepth])) != hipSuccess) return e;
    const Tile t = tile(c);
    const float* src = ring_[c % kDepth];
    const size_t bytes = t.nr * t.nc * sizeof(float);
    // pieces of whole rows; a tile of one long row is cut along the row instead
    const bool by_cols = t.nr == 1;
    const size_t units = by_cols ? (t.nc + 4095) / 4096 : t.nr;
    if (widen) {
      double* out = static_cast<double*>(dst) + t.r0 * dst_pitch + t.c0;
      Workers::get().run(units, worker_count(bytes), [=](size_t a, size_t b) {
        if (by_cols) { widen_row(src + a * 4096, out + a * 4096, std::min(t.nc, b * 4096) - a * 4096); return; }
        for (size_t r = a; r < b; ++r) widen_row(src + r * t.nc, out + r * dst_pitch, t.nc);
      });
    } else {
      float* out = static_cast<float*>(dst) + t.r0 * dst_pitch + t.c0;
      Workers::get().run(units, worker_count(bytes), [=](size_t a, size_t b) {
        if (by_cols) { std::memcpy(out + a * 4096, src + a * 4096, (std::min(t.nc, b * 4096) - a * 4096) * sizeof(float)); return; }
        for (size_t r = a; r < b; ++r) std::memcpy(out + r * dst_pitch, src + r * t.nc, t.nc * sizeof(float));
      });
    }
  }
  return hipStreamSynchronize(stream);
}

}  // namespace gcwt
