// Issue rate of the vector instructions k_synth7 is made of, at the occupancy it runs at
// (4 waves per SIMD) and at 1 wave per SIMD: cycles per wave-instruction per SIMD, from
// in-kernel s_memtime.  Answers whether v_pk_*_f32 costs one or two issue slots on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

enum { PK_ADD, PK_MUL, PK_FMA, PK_ADD_MOD, ADD, FMA, MUL, SQRT, MIX_PK, MIX_SC, PK_FMA_BC, PK_FMA_S2, PK_FMA_S1, FMA_S, N_OPS };
static const char* kNames[] = {"v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32 op_sel/neg",
                               "v_add_f32", "v_fma_f32", "v_mul_f32", "v_sqrt_f32",
                               "cmul packed (pk_mul+pk_fma)", "cmul scalar (2 mul + 2 fma)",
                               "v_pk_fma_f32 src0 broadcast", "v_pk_fma_f32 src1 = SGPR pair (both)",
                               "v_pk_fma_f32 src1 = SGPR (one dword)", "v_fma_f32 src1 = SGPR"};

template <int OP>
__global__ void __launch_bounds__(512) k_rate(float* out, long long* cyc, int iters) {
  v2f a[8];
  const v2f w = {1.0001f + threadIdx.x * 1e-7f, 0.9999f};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (v2f){1.f + i, 2.f + threadIdx.x * 1e-3f};
  const v2f ws = {1e-9f * iters, 1e-9f};     // wave-uniform: lives in SGPRs
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
        else if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
        else if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
        else if (OP == PK_ADD_MOD) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(w));
        else if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
        else if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(w.x));
        else if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
        else if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
        else if (OP == PK_FMA_BC) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(w));
        else if (OP == PK_FMA_S2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(ws));
        else if (OP == PK_FMA_S1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(ws));
        else if (OP == FMA_S) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(a[(i + 1) & 7].y), "s"(ws.x));
        else if (OP == MIX_PK) {
          v2f t;
          asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a[i]), "v"(w));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(a[i]) : "v"(a[i]), "v"(w), "v"(t));
        } else if (OP == MIX_SC) {
          float tr, ti;
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tr) : "v"(a[i].x), "v"(w.x));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ti) : "v"(a[i].x), "v"(w.y));
          asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(a[i].x) : "v"(a[i].y), "v"(w.y), "v"(tr));
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i].y) : "v"(a[i].y), "v"(w.x), "v"(ti));
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(int threads, int blocks_per_cu, float* out, long long* cyc, std::vector<long long>& h) {
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(threads), 0, 0, out, cyc, 10);
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int waves = grid * threads / 64;
  CK(hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.begin() + waves);
  const double med = (double)h[waves / 2];
  const int per_iter = (OP == MIX_PK ? 2 : OP == MIX_SC ? 4 : 1) * 32;
  const double waves_per_simd = (double)threads / 64 * blocks_per_cu / 4;
  printf("%-30s %d thr x %d/CU (%.0f waves/SIMD): %7.2f cyc per wave-instr in the wave, %6.2f cyc/instr/SIMD, %.3f ms\n",
         kNames[OP], threads, blocks_per_cu, waves_per_simd, med / ((double)iters * per_iter),
         med / ((double)iters * per_iter) / waves_per_simd, ms);
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, sizeof(long long) * 65536));
  std::vector<long long> h(65536);
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int thr = cfg == 0 ? 256 : 512, bpc = cfg == 0 ? 1 : 2;
    run<PK_ADD>(thr, bpc, out, cyc, h); run<PK_MUL>(thr, bpc, out, cyc, h); run<PK_FMA>(thr, bpc, out, cyc, h);
    run<PK_ADD_MOD>(thr, bpc, out, cyc, h); run<ADD>(thr, bpc, out, cyc, h); run<FMA>(thr, bpc, out, cyc, h);
    run<MUL>(thr, bpc, out, cyc, h); run<SQRT>(thr, bpc, out, cyc, h); run<MIX_PK>(thr, bpc, out, cyc, h);
    run<MIX_SC>(thr, bpc, out, cyc, h);
    run<PK_FMA_BC>(thr, bpc, out, cyc, h); run<PK_FMA_S2>(thr, bpc, out, cyc, h); run<PK_FMA_S1>(thr, bpc, out, cyc, h);
    run<FMA_S>(thr, bpc, out, cyc, h);
  }
  return 0;
}
