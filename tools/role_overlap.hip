// Can the LDS pipe and the VALU work at the same time when DIFFERENT waves of a CU use
// them (role-specialised waves), or do LDS and VALU time add whatever issues them?
// 512-thread workgroups, 2 per CU: waves 0-3 run an LDS-only loop, waves 4-7 a VALU-only
// loop (one of each per SIMD and workgroup); compared with each half running alone.
//   hipcc -O3 --offload-arch=gfx950 tools/role_overlap.hip -o /tmp/ro && /tmp/ro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

// mode bit 0: waves 0-3 do LDS work; bit 1: waves 4-7 do VALU work
__global__ void __launch_bounds__(512) k_roles(float* out, long long* cyc, int iters, int mode) {
  __shared__ __attribute__((aligned(16))) v2f ex[8192];
  v2f a[16];
  const v2f w = {1.0001f, 0.9999f};
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (v2f){1.f + i, 2.f + threadIdx.x * 1e-3f};
  for (int i = threadIdx.x; i < 8192; i += 512) ex[i] = (v2f){0.f, 0.f};
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  v2f* const p = ex + (threadIdx.x & 255);
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) p[256 * j] = a[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = p[256 * ((j + 1) & 15)];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const v2f q = p[256 * j + 4096]; a[j] += q; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
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
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

void run(const char* name, int mode, float* out, long long* cyc, std::vector<long long>& h) {
  const int iters = 1000, grid = 512;
  hipLaunchKernelGGL(k_roles, dim3(grid), dim3(512), 0, 0, out, cyc, 5, mode);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_roles, dim3(grid), dim3(512), 0, 0, out, cyc, iters, mode);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h.data(), cyc, sizeof(long long) * grid * 8, hipMemcpyDeviceToHost));
  std::vector<long long> lds, valu;
  for (int b = 0; b < grid; ++b) for (int wv = 0; wv < 8; ++wv) (wv < 4 ? lds : valu).push_back(h[b * 8 + wv]);
  std::sort(lds.begin(), lds.end()); std::sort(valu.begin(), valu.end());
  printf("%-28s LDS waves %7.0f cycles/iter, VALU waves %7.0f cycles/iter, kernel %.3f ms\n", name,
         (double)lds[lds.size() / 2] / iters, (double)valu[valu.size() / 2] / iters, ms);
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, sizeof(long long) * 65536));
  std::vector<long long> h(65536);
  run("LDS waves alone", 1, out, cyc, h);
  run("VALU waves alone", 2, out, cyc, h);
  run("both roles together", 3, out, cyc, h);
  return 0;
}
