// Dependent-issue latency of the VALU instructions k_synth7 uses: one wave per SIMD, every
// instruction consuming the previous one's result, and the same with 2 and 4 independent
// chains (how much instruction-level parallelism one wave needs to issue back to back).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_latency.hip -o /tmp/vlat && /tmp/vlat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
enum { PK_ADD, PK_FMA, ADD, FMA, N_OPS };
static const char* kNames[] = {"v_pk_add_f32", "v_pk_fma_f32", "v_add_f32", "v_fma_f32"};

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k_lat(float* out, long long* cyc, int iters) {
  v2f a[CHAINS];
  const v2f w = {1.0001f, 0.9999f};
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) a[i] = (v2f){1.f + i, 2.f + threadIdx.x * 1e-3f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 32 / CHAINS; ++rep)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
        else if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
        else if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP, int CHAINS>
void run(float* out, long long* cyc, std::vector<long long>& h, int wps) {
  const int iters = 2000, threads = wps == 1 ? 256 : 512, grid = wps == 4 ? 512 : 256;
  hipLaunchKernelGGL((k_lat<OP, CHAINS>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  const int waves = grid * threads / 64;
  CK(hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.begin() + waves);
  const double per = (double)h[waves / 2] / (iters * 32.0);
  printf("%-14s %d chain(s), %d wave(s)/SIMD: %6.2f cycles per instruction in the wave, %5.2f per SIMD\n", kNames[OP], CHAINS,
         wps, per, per / wps);
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, sizeof(long long) * 65536));
  std::vector<long long> h(65536);
  for (int wps : {1, 2, 4}) {
    run<PK_ADD, 1>(out, cyc, h, wps); run<PK_ADD, 2>(out, cyc, h, wps); run<PK_ADD, 4>(out, cyc, h, wps);
    run<PK_FMA, 1>(out, cyc, h, wps); run<PK_FMA, 2>(out, cyc, h, wps);
    run<ADD, 1>(out, cyc, h, wps); run<ADD, 2>(out, cyc, h, wps); run<FMA, 1>(out, cyc, h, wps);
  }
  return 0;
}
