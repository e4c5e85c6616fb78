// host_out.h -- moving a finished result from the device into the caller's (pageable) host
// array: tiles of rows go through a ring of four pinned staging buffers; while the next
// tiles are on the wire, a standing pool of worker threads scatters tile i into the destination,
// widening float32 to float64 (streaming stores) when the caller asked for the reference's dtype
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
  // rows dst_pitch elements apart (0: dense), as float32 or (widen) float64.  Blocks until dst is
  // complete.  stream: the stream the producer kernels ran on.
  hipError_t drain(const float* d_src, size_t src_pitch_floats, size_t n_rows, size_t row_floats,
                   void* dst, bool widen, hipStream_t stream, size_t dst_pitch = 0);
  void release();

 private:
  static constexpr int kDepth = 4;
  static constexpr size_t kChunkBytes = size_t(16) << 20;
  float* ring_[kDepth] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t landed_[kDepth] = {nullptr, nullptr, nullptr, nullptr};
};

}  // namespace gcwt
