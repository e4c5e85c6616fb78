// Do LDS instructions and packed-f32 VALU instructions of co-resident waves overlap on
// gfx950, or do their times add?  One loop body with k_synth7's per-batch mix (192 v_pk,
// 16 ds_write_b64, 32 ds_read_b64, 16 ds_read_b32), run as VALU only, LDS only and both,
// at 4 waves per SIMD.  Cycles from s_memtime around the whole loop.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_lds_overlap.hip -o /tmp/vlo && /tmp/vlo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool VALU, bool LDS, int WIDE>
__global__ void __launch_bounds__(512) k_mix(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) v2f ex[16 * 513 + 512];
  v2f a[16];
  const v2f w = {1.0001f, 0.9999f};
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (v2f){1.f + i, 2.f + threadIdx.x * 1e-3f};
  v2f* const wr = ex + (threadIdx.x & 15) * 513 + (threadIdx.x >> 4);
  const v2f* const rd = ex + threadIdx.x;
  const float* const g = reinterpret_cast<const float*>(ex) + (threadIdx.x & 15);
  for (int i = threadIdx.x; i < 16 * 513 + 512; i += 512) ex[i] = (v2f){0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { const float q = g[16 * j]; a[j].x += q; }   // gain reads (b32)
    }
    if (VALU) {
#pragma unroll
      for (int rep = 0; rep < 6; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(a[(i + 1) & 15]));
    }
    if (LDS) {
      if (WIDE) {
#pragma unroll
        for (int j = 0; j < 16; j += 2) {   // twiddle reads as b128
          const float4 q = *reinterpret_cast<const float4*>(ex + 16 * 513 + 2 * ((16 * j + (threadIdx.x & 15)) >> 1));
          a[j].x += q.x; a[j + 1].x += q.z;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) { const v2f q = ex[16 * 513 + 16 * j + (threadIdx.x & 15)]; a[j] += q; }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) wr[j * 32] = a[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = rd[j * 513];
      __syncthreads();
    }
    if (VALU) {
#pragma unroll
      for (int rep = 0; rep < 6; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(a[(i + 1) & 15]));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <bool VALU, bool LDS, int WIDE>
void run(const char* name, float* out, long long* cyc, std::vector<long long>& h) {
  const int iters = 400, grid = 512;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mix<VALU, LDS, WIDE>), dim3(grid), dim3(512), 0, 0, out, cyc, 5);
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_mix<VALU, LDS, WIDE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int waves = grid * 8;
  CK(hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.begin() + waves);
  printf("%-40s %8.0f cycles per iteration per wave (4 waves/SIMD), %.3f ms, clock %.2f GHz\n", name,
         (double)h[waves / 2] / iters, ms, (double)h[waves / 2] / (ms * 1e6));
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, sizeof(long long) * 65536));
  std::vector<long long> h(65536);
  run<true, false, 0>("VALU only (192 v_pk_fma)", out, cyc, h);
  run<false, true, 0>("LDS only (16 w64, 32 r64, 16 r32, 2 bar)", out, cyc, h);
  run<true, true, 0>("both", out, cyc, h);
  run<false, true, 1>("LDS only, twiddles as b128", out, cyc, h);
  run<true, true, 1>("both, twiddles as b128", out, cyc, h);
  return 0;
}
