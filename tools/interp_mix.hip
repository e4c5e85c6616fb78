// Prices the inner loop of an interpolating synthesis (round 3): instead of one point of a
// 256-point inverse FFT per stored sample, a scale's complex output is made at q x its band
// and brought to the full rate by a T-tap polyphase FIR with real coefficients,
//     y[I m' + rho] = sum_j c_rho[j] z[m' + j - T/2 + 1],
// then |.| and store.  One lane-task = NOUT consecutive output samples (one interval of the
// coarse signal: the window of T complex values comes from LDS once per task, the NOUT x T
// coefficients live in registers), a wave = 64 consecutive tasks.
// Runs the loop chip-wide (2 x 512 threads per CU) for a few seconds so that rocm-smi can be
// sampled beside it (tools/interp_mix.sh) and prints the rate: compare with `mixstore` of
// tools/power_mix.hip (the instruction mix of k_synth7 with its 14 stores per scale).
//   modes: fir8   T = 8, 8 outputs per lane, two 16-byte stores at a lane stride of 32 B
//          fir8c  T = 8, 4 outputs per lane, one 16-byte store, 1 KB contiguous per wave store
//          fir6   T = 6, 8 outputs per lane
//          fir6c / fir4c  fir8c with T = 6 / 4
//          fir8ns / fir6ns  the same without stores
//          st32   only the stores of fir8;  st16  only the stores of fir8c
//   hipcc -O3 --offload-arch=gfx950 tools/interp_mix.hip -o /tmp/imix && /tmp/imix fir8 5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { FIR8, FIR8C, FIR6, FIR8NS, FIR6NS, ST32, ST16, FIR6C, FIR4C, MFMA8, MFMA8NS, ST4, N_MODES };
static const char* kNames[N_MODES] = {"fir8", "fir8c", "fir6", "fir8ns", "fir6ns", "st32", "st16", "fir6c", "fir4c", "mfma8", "mfma8ns", "st4"};

// acc += z * c.x  /  acc += z * c.y   (complex z, real coefficient broadcast to both halves)
__device__ __forceinline__ void fma_lo(v2f& acc, v2f z, v2f c) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(z), "v"(c));
}
__device__ __forceinline__ void fma_hi(v2f& acc, v2f z, v2f c) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(z), "v"(c));
}
__device__ __forceinline__ v2f mul_lo(v2f z, v2f c) {
  v2f r;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(z), "v"(c));
  return r;
}

constexpr int kZ = 8192;   // complex samples of the coarse signal parked in LDS (64 KB)

template <int MODE>
__global__ void __launch_bounds__(512, 4) k_run(float* out, float* sink, int iters, long long* clk) {
  constexpr int T = (MODE == FIR6 || MODE == FIR6NS || MODE == FIR6C) ? 6 : MODE == FIR4C ? 4 : 8;
  constexpr int NOUT = MODE == FIR8C || MODE == ST16 || MODE == FIR6C || MODE == FIR4C ? 4 : 8;
  constexpr bool kStore = !(MODE == FIR8NS || MODE == FIR6NS);
  constexpr bool kFir = !(MODE == ST32 || MODE == ST16);
  __shared__ __attribute__((aligned(16))) v2f z[kZ + 16];
  const int tid = threadIdx.x;
  for (int i = tid; i < kZ + 16; i += 512) z[i] = (v2f){1e-3f * (i & 63), 1e-3f};
  // coefficient pairs (c[i][2jj], c[i][2jj+1]) of this lane's NOUT output positions
  v2f c[NOUT][T / 2];
#pragma unroll
  for (int i = 0; i < NOUT; ++i)
#pragma unroll
    for (int j = 0; j < T / 2; ++j) c[i][j] = (v2f){0.1f + 0.01f * i + 1e-3f * (tid & 1), 0.12f - 0.01f * j};
  // each workgroup streams through its own 4 MB of the sink: 512 tasks x 32 B = 16 KB per iteration
  char* const base = reinterpret_cast<char*>(sink) + (size_t)blockIdx.x * (4u << 20);
  float acc_keep = 0.f;
  __syncthreads();
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    // interval of this task: consecutive lanes, consecutive intervals (I = 8: one task per interval)
    constexpr int kTasks = NOUT == 8 ? 1 : 2;
#pragma unroll
    for (int task = 0; task < kTasks; ++task) {
      const int m = ((it * 512 + tid) * 1 + task * 256) & (kZ - 1);
      float r[NOUT];
      if (kFir) {
        v2f w[T];
#pragma unroll
        for (int j = 0; j < T; ++j) w[j] = z[m + j];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
          v2f a = mul_lo(w[0], c[i][0]);
          fma_hi(a, w[1], c[i][0]);
#pragma unroll
          for (int j = 1; j < T / 2; ++j) { fma_lo(a, w[2 * j], c[i][j]); fma_hi(a, w[2 * j + 1], c[i][j]); }
          r[i] = __builtin_amdgcn_sqrtf(__builtin_fmaf(a.y, a.y, a.x * a.x));
        }
      } else {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) r[i] = 1.f + i + tid;
      }
      if (kStore) {
        const size_t off = ((size_t)(it & 255) * 512 + tid) * 32;
        if (NOUT == 8) {
          __builtin_nontemporal_store((v4f){r[0], r[1], r[2], r[3]}, reinterpret_cast<v4f*>(base + off));
          __builtin_nontemporal_store((v4f){r[4], r[5], r[6], r[7]}, reinterpret_cast<v4f*>(base + off + 16));
        } else {
          const size_t o4 = ((size_t)(it & 255) * 1024 + task * 512 + tid) * 16;
          __builtin_nontemporal_store((v4f){r[0], r[1], r[2], r[3]}, reinterpret_cast<v4f*>(base + o4));
        }
      } else {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) acc_keep += r[i];
      }
    }
  }
  const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (acc_keep == 123.456f) out[0] = acc_keep;
  if (blockIdx.x == 7 && tid == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

// Round 4: the same 8-tap polyphase FIR on the matrix pipe.  Out[m][rho] = sum_j z[m + j] c_rho[j] for 32 intervals m
// x 32 sub-sample positions rho is a 32 x 8 (Hankel of z) by 8 x 32 (coefficients) product: four
// v_mfma_f32_32x32x2_f32 for the real parts and four for the imaginary parts per 1024 outputs of a wave (full fp32
// arithmetic; the fp32 matrix rate equals the packed-vector rate, but it is another pipe: the vector pipe keeps |.|,
// the LDS reads and the stores).  A lane ends up with column rho = lane % 32 and 16 intervals: a store instruction
// writes two 128-byte runs (dword stores), like k_synth7.
//   mfma8   with stores    mfma8ns  without    st4  only those stores
typedef float v16f __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(512, 4) k_run_mfma(float* out, float* sink, int iters, long long* clk) {
  constexpr bool kStore = MODE != MFMA8NS;
  constexpr bool kFir = MODE != ST4;
  __shared__ __attribute__((aligned(16))) v2f z[kZ + 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kZ + 64; i += 512) z[i] = (v2f){1e-3f * (i & 63), 1e-3f};
  // B operand: coefficient c[rho][2 i + k], rho = lane % 32, k = lane / 32, one register per instruction
  float cb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cb[i] = 0.1f + 0.01f * (lane & 31) - 0.01f * (2 * i + (lane >> 5));
  char* const base = reinterpret_cast<char*>(sink) + (size_t)blockIdx.x * (4u << 20);
  float acc_keep = 0.f;
  __syncthreads();
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    // this wave's 32 intervals of the iteration: 8 waves x 32 = 256 intervals x 32 positions = 8192 outputs = 32 KB
    const int m0 = ((it * 8 + wave) * 32) & (kZ - 1);
    v16f are, aim;
#pragma unroll
    for (int v = 0; v < 16; ++v) { are[v] = 0.f; aim[v] = 0.f; }
    if (kFir) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v2f a = z[m0 + (lane & 31) + 2 * i + (lane >> 5)];         // A[m][k] = z[m + 2 i + k]
        are = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, cb[i], are, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, cb[i], aim, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int v = 0; v < 16; ++v) are[v] = 1.f + v + tid;
    }
    // lane: column rho = lane % 32, rows (intervals) 8 (v / 4) + 4 (lane / 32) + v % 4; sample = 32 interval + rho
    float* const dst = reinterpret_cast<float*>(base + ((size_t)(it & 127) * 8 + wave) * 4096);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float r = kFir ? __builtin_amdgcn_sqrtf(__builtin_fmaf(aim[v], aim[v], are[v] * are[v])) : are[v];
      const int row = 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
      if (kStore) __builtin_nontemporal_store(r, dst + 32 * row + (lane & 31));
      else acc_keep += r;
    }
  }
  const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (acc_keep == 123.456f) out[0] = acc_keep;
  if (blockIdx.x == 7 && tid == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int MODE>
void drive_mfma(double seconds, float* out, float* sink, long long* clk, int grid) {
  const int iters = 2048;
  hipLaunchKernelGGL((k_run_mfma<MODE>), dim3(grid), dim3(512), 0, 0, out, sink, 10, clk);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  do {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((k_run_mfma<MODE>), dim3(grid), dim3(512), 0, 0, out, sink, iters, clk);
    CK(hipDeviceSynchronize());
    launches += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } while (el < seconds);
  long long h[2];
  CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double per_iter_us = el * 1e6 / ((double)launches * iters);
  const double bytes = (double)grid * 8 * 4096;          // 8 waves x 1024 outputs x 4 B per workgroup and iteration
  printf("%-7s %.2f s, %.3f us per iteration (16 samples per lane), clock %.3f GHz, %.2f TB/s\n",
         kNames[MODE], el, per_iter_us, (double)h[0] / ((double)h[1] * 10.0), bytes / per_iter_us / 1e6);
}

template <int MODE>
void drive(double seconds, float* out, float* sink, long long* clk, int grid) {
  const int iters = 2048;
  hipLaunchKernelGGL((k_run<MODE>), dim3(grid), dim3(512), 0, 0, out, sink, 10, clk);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  do {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((k_run<MODE>), dim3(grid), dim3(512), 0, 0, out, sink, iters, clk);
    CK(hipDeviceSynchronize());
    launches += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } while (el < seconds);
  long long h[2];
  CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double per_iter_us = el * 1e6 / ((double)launches * iters);
  // 8 outputs per thread per iteration: 512 x 8 x 4 B per workgroup
  const double bytes = (double)grid * 512 * 32;
  printf("%-7s %.2f s, %.3f us per iteration (8 samples per lane; %.3f us per 14 wave-rows of a CU's 16 waves), clock %.3f GHz, %.2f TB/s\n",
         kNames[MODE], el, per_iter_us, per_iter_us * 14.0 / 8.0 * (512.0 / grid), (double)h[0] / ((double)h[1] * 10.0), bytes / per_iter_us / 1e6);
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "fir8";
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  float *out, *sink; long long* clk;
  const int grid = 512;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, 64));
  CK(hipMalloc(&sink, (size_t)grid * (4u << 20)));
  if (!strcmp(mode, "idle")) { printf("idle\n"); fflush(stdout); std::this_thread::sleep_for(std::chrono::duration<double>(seconds)); return 0; }
#define CASE(M) if (!strcmp(mode, kNames[M])) { drive<M>(seconds, out, sink, clk, grid); return 0; }
  CASE(FIR8) CASE(FIR8C) CASE(FIR6) CASE(FIR8NS) CASE(FIR6NS) CASE(ST32) CASE(ST16) CASE(FIR6C) CASE(FIR4C)
#define CASEM(M) if (!strcmp(mode, kNames[M])) { drive_mfma<M>(seconds, out, sink, clk, grid); return 0; }
  CASEM(MFMA8) CASEM(MFMA8NS) CASEM(ST4)
  printf("unknown mode %s\n", mode);
  return 1;
}
