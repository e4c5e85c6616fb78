// Would two wave groups of one workgroup, held in OPPOSITE phases by workgroup-wide
// barriers, overlap the LDS exchange of one with the arithmetic of the other?  Same per-scale
// instruction mix as tools/valu_lds_overlap.hip (k_synth7's), two layouts:
//   alike    : 2 workgroups x 512 threads per CU, every wave runs A, W, barrier, R, barrier, B
//   pingpong : 1 workgroup x 1024 threads per CU, groups of 8 waves, 4 barriers per scale:
//              group 0:  B | A | W | R          group 1:  W | R | B | A
// Cycles from s_memtime around the loop; the same total work in both layouts.
//   hipcc -O3 --offload-arch=gfx950 tools/pingpong_overlap.hip -o /tmp/ppo && /tmp/ppo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kPlane = 513;
constexpr int kGroupLds = 16 * kPlane + 512;   // exchange planes + twiddles (+ gains share the front)

// WHAT bit 0: the arithmetic, bit 1: the LDS traffic
template <int WHAT = 3>
__device__ __forceinline__ void stage_a(v2f (&a)[16], const v2f w, const v2f* ex, int t) {
  const float* const g = reinterpret_cast<const float*>(ex) + t;
  if (WHAT & 2) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float q = g[16 * j]; a[j].x += q; }          // gains (b32)
  }
  if (WHAT & 1) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(a[(i + 1) & 15]));
  }
  if (WHAT & 2) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { const v2f q = ex[16 * kPlane + 16 * j + t]; a[j] += q; }   // twiddles (b64)
  }
}
template <int WHAT = 3>
__device__ __forceinline__ void stage_b(v2f (&a)[16], const v2f w) {
  if (WHAT & 1) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(a[(i + 1) & 15]));
  }
}

template <int STORES>
__global__ void __launch_bounds__(512) k_alike(float* out, long long* cyc, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) v2f ex[];
  v2f a[16];
  const v2f w = {1.0001f, 0.9999f};
  const int tid = threadIdx.x, t = tid & 15;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (v2f){1.f + i, 2.f + tid * 1e-3f};
  v2f* const wr = ex + t * kPlane + (tid >> 4);
  const v2f* const rd = ex + tid;
  for (int i = tid; i < kGroupLds; i += 512) ex[i] = (v2f){0.f, 0.f};
  float* const dst = sink + ((size_t)blockIdx.x * 512 + tid);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    stage_a(a, w, ex, t);
#pragma unroll
    for (int j = 0; j < 16; ++j) wr[j * 32] = a[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = rd[j * kPlane];
    __syncthreads();
    stage_b(a, w);
    if (STORES) {
#pragma unroll
      for (int j = 1; j < 15; ++j) __builtin_nontemporal_store(a[j].x, dst + (size_t)((it * 14 + j) & 1023) * (size_t)gridDim.x * 512);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) out[0] = s;
  if ((tid & 63) == 0) cyc[(blockIdx.x * blockDim.x + tid) >> 6] = t1 - t0;
}

template <int STORES, int WHAT = 3>   // WHAT bit 0: arithmetic, bit 1: LDS traffic
__global__ void __launch_bounds__(1024) k_pingpong(float* out, long long* cyc, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  v2f a[16];
  const v2f w = {1.0001f, 0.9999f};
  const int grp = threadIdx.x >> 9, tid = threadIdx.x & 511, t = tid & 15;
  v2f* const ex = lds + grp * kGroupLds;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (v2f){1.f + i, 2.f + tid * 1e-3f};
  v2f* const wr = ex + t * kPlane + (tid >> 4);
  const v2f* const rd = ex + tid;
  for (int i = threadIdx.x; i < 2 * kGroupLds; i += 1024) lds[i] = (v2f){0.f, 0.f};
  float* const dst = sink + ((size_t)blockIdx.x * 1024 + threadIdx.x);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (grp == 0) {
    stage_a<WHAT>(a, w, ex, t);           // group 0 runs half a scale ahead
    for (int it = 0; it < iters; ++it) {
      if (WHAT & 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) wr[j * 32] = a[j];
      }
      __syncthreads();
      if (WHAT & 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = rd[j * kPlane];
      }
      __syncthreads();
      stage_b<WHAT>(a, w);
      if (STORES) {
#pragma unroll
        for (int j = 1; j < 15; ++j) __builtin_nontemporal_store(a[j].x, dst + (size_t)((it * 14 + j) & 1023) * (size_t)gridDim.x * 1024);
      }
      __syncthreads();
      stage_a<WHAT>(a, w, ex, t);
      __syncthreads();
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (it > 0) {
        stage_b<WHAT>(a, w);
        if (STORES) {
#pragma unroll
          for (int j = 1; j < 15; ++j) __builtin_nontemporal_store(a[j].x, dst + (size_t)((it * 14 + j) & 1023) * (size_t)gridDim.x * 1024);
        }
      }
      __syncthreads();
      stage_a<WHAT>(a, w, ex, t);
      __syncthreads();
      if (WHAT & 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) wr[j * 32] = a[j];
      }
      __syncthreads();
      if (WHAT & 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = rd[j * kPlane];
      }
      __syncthreads();
    }
    stage_b<WHAT>(a, w);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int grid, int threads, size_t lds, float* out, long long* cyc, float* sink, std::vector<long long>& h) {
  const int iters = 400;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, out, cyc, 5, sink);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, out, cyc, iters, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int waves = grid * threads / 64;
  CK(hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.begin() + waves);
  printf("%-34s %8.0f cycles per scale per wave, %.3f ms for %d wave-scales -> %.2f ns per CU per 16 wave-scales\n", name,
         (double)h[waves / 2] / iters, ms, waves * iters, ms * 1e6 / ((double)waves * iters / 16.0 / 256.0));
}

int main() {
  float *out, *sink; long long* cyc;
  const int cus = 256;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, sizeof(long long) * 65536));
  CK(hipMalloc(&sink, (size_t)1024 * cus * 2 * 512 * 4));   // 1 GiB: 1024 rows of one float per thread
  std::vector<long long> h(65536);
  const size_t lds1 = sizeof(v2f) * kGroupLds, lds2 = 2 * lds1;
  run("alike, no stores", k_alike<0>, 2 * cus, 512, lds1, out, cyc, sink, h);
  run("pingpong, no stores", k_pingpong<0, 3>, cus, 1024, lds2, out, cyc, sink, h);
  run("pingpong, arithmetic only", k_pingpong<0, 1>, cus, 1024, lds2, out, cyc, sink, h);
  run("pingpong, LDS traffic only", k_pingpong<0, 2>, cus, 1024, lds2, out, cyc, sink, h);
  run("pingpong, barriers only", k_pingpong<0, 0>, cus, 1024, lds2, out, cyc, sink, h);
  run("alike, 14 stores per scale", k_alike<1>, 2 * cus, 512, lds1, out, cyc, sink, h);
  run("pingpong, 14 stores per scale", k_pingpong<1, 3>, cus, 1024, lds2, out, cyc, sink, h);
  return 0;
}
