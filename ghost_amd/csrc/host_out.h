// host_out.h -- moving a finished result from the device into the caller's (pageable) host
// array: chunks of rows go through a ring of two pinned staging buffers; while chunk i+1
// is on the wire, worker threads scatter chunk i into the destination, widening float32
// to float64 on the way when the caller asked for the reference's dtype
// (ghost/wave/transforms.py:185 allocates float64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace gcwt {

class HostOut {
 public:
  HostOut() = default;
  ~HostOut();
  HostOut(const HostOut&) = delete;
  HostOut& operator=(const HostOut&) = delete;
  // Copies n_rows rows of row_floats float32 each (device pitch src_pitch_floats) into dst,
  // dense, as float32 or (widen) float64.  Blocks until dst is complete.  stream: the
  // stream the producer kernels ran on.
  hipError_t drain(const float* d_src, size_t src_pitch_floats, size_t n_rows, size_t row_floats,
                   void* dst, bool widen, hipStream_t stream);
  void release();

 private:
  static constexpr size_t kChunkBytes = size_t(32) << 20;
  float* ring_[2] = {nullptr, nullptr};
  hipEvent_t landed_[2] = {nullptr, nullptr};
};

}  // namespace gcwt
