// LDS pipe cost per wave-instruction on gfx950 with 16 waves per CU all issuing the same
// DS op (conflict-free addresses): cycles per instruction per CU and bytes per clock.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_rate.hip -o /tmp/lds_rate && /tmp/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { R32, R32B, R64, R64B, R128, R128B, W32, W64, W128, W2X64, W2ST64X64, R2X64, N_OPS };
static const char* kNames[] = {"ds_read_b32 (lane-contiguous)", "ds_read_b32 (16 distinct addresses per wave)",
  "ds_read_b64 (lane-contiguous)", "ds_read_b64 (16 distinct addresses per wave)",
  "ds_read_b128 (lane-contiguous)", "ds_read_b128 (16 distinct addresses per wave)",
  "ds_write_b32", "ds_write_b64", "ds_write_b128",
  "ds_write2_b64 (two 8-byte values, 256 B apart)", "ds_write2st64_b64 (two 8-byte values, 4 KB apart)",
  "ds_read2_b64 (two 8-byte values, 256 B apart)"};
static const int kBytes[] = {4, 4, 8, 8, 16, 16, 4, 8, 16, 16, 16, 16};

template <int OP>
__global__ void __launch_bounds__(512) k_lds(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) buf[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int elem = OP >= W2X64 ? 2 : kBytes[OP] / 4;
  const bool bc = OP == R32B || OP == R64B || OP == R128B;
  // byte address of this lane's access; each of the 8 unrolled accesses adds 2 KB
  unsigned addr = (unsigned)(((bc ? (lane & 15) : lane) * elem + w * 64 * elem) * 4) % 16384u;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned a = (addr + 2048u * u) & 65535u;
      if (OP == R32 || OP == R32B) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); acc0 += v; }
      else if (OP == R64 || OP == R64B) { f2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); acc0 += v.x; acc1 += v.y; }
      else if (OP == R128 || OP == R128B) { f4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); acc0 += v.x; acc1 += v.y; acc2 += v.z; acc3 += v.w; }
      else if (OP == W32) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(acc0) : "memory");
      else if (OP == W64) { f2 v = {acc0, acc1}; asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory"); }
      else if (OP == W128) { f4 v = {acc0, acc1, acc2, acc3}; asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory"); }
      else if (OP == W2X64) { f2 v = {acc0, acc1}, z = {acc2, acc3}; asm volatile("ds_write2_b64 %0, %1, %2 offset1:32" :: "v"(a & 32767u), "v"(v), "v"(z) : "memory"); }
      else if (OP == W2ST64X64) { f2 v = {acc0, acc1}, z = {acc2, acc3}; asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:8" :: "v"(a & 32767u), "v"(v), "v"(z) : "memory"); }
      else { f4 v; asm volatile("ds_read2_b64 %0, %1 offset1:32" : "=v"(v) : "v"(a & 32767u)); acc0 += v.x; acc1 += v.y; acc2 += v.z; acc3 += v.w; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (acc0 + acc1 + acc2 + acc3 == 123.456f) out[0] = acc0;
  if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(float* out, long long* cyc, std::vector<long long>& h) {
  const int iters = 2000, grid = 512;
  hipLaunchKernelGGL(k_lds<OP>, dim3(grid), dim3(512), 0, 0, out, cyc, 5);
  hipLaunchKernelGGL(k_lds<OP>, dim3(grid), dim3(512), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  const int waves = grid * 8;
  CK(hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.begin() + waves);
  const double per_wave_instr = (double)h[waves / 2] / (iters * 8.0);   // cycles per instr as one wave sees it
  const double per_cu = per_wave_instr / 16.0;                         // 16 waves share the CU's LDS
  printf("%-48s %6.2f cycles per wave-instr per CU, %6.1f B/clk/CU\n", kNames[OP], per_cu, 64.0 * kBytes[OP] / per_cu);
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, sizeof(long long) * 65536));
  std::vector<long long> h(65536);
  run<R32>(out, cyc, h); run<R32B>(out, cyc, h); run<R64>(out, cyc, h); run<R64B>(out, cyc, h);
  run<R128>(out, cyc, h); run<R128B>(out, cyc, h); run<W32>(out, cyc, h); run<W64>(out, cyc, h); run<W128>(out, cyc, h);
  run<W2X64>(out, cyc, h); run<W2ST64X64>(out, cyc, h); run<R2X64>(out, cyc, h);
  return 0;
}
