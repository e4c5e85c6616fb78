// What does the package draw for each ingredient of k_synth7's loop?  Runs one ingredient
// chip-wide (2 x 512 threads per CU, 4 waves per SIMD) for a few seconds so that
// `rocm-smi --showpower --showclocks` can be sampled beside it (tools/power_mix.sh), and
// prints the rate it reached.  Modes: pk_fma, pk_add, sqrt, lds_w64, lds_r64, lds_r128, exchange
// (16 w64 + barrier + 16 r64 + barrier), dpp_xpose (the same 16 x 16 transpose over the lanes of a DPP
// row, no LDS), store (nt dword stores, 256 B per wave), mix (the
// kernel's per-scale mix without stores), mixstore (with 14 stores per scale).
//   hipcc -O3 --offload-arch=gfx950 tools/power_mix.hip -o /tmp/pmix && /tmp/pmix pk_fma 5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { PK_FMA, PK_ADD, SQRT, LDS_W64, LDS_R64, LDS_R128, EXCHANGE, STORE, MIX, MIXSTORE, STORE_DEF, STORE4, STORE4_DEF, STORE2, DPP_ADD, ADD, DPP_XPOSE, N_MODES };
static const char* kNames[N_MODES] = {"pk_fma", "pk_add", "sqrt", "lds_w64", "lds_r64", "lds_r128", "exchange", "store", "mix", "mixstore", "store_def", "store4", "store4_def", "store2", "dpp_add", "add", "dpp_xpose"};

constexpr int kPlane = 513;

template <int MODE>
__global__ void __launch_bounds__(512) k_run(float* out, float* sink, int iters, long long* clk) {
  __shared__ __attribute__((aligned(16))) v2f ex[16 * kPlane + 512];
  v2f a[16];
  const v2f w = {1.0001f, 0.9999f};
  const int tid = threadIdx.x, t = tid & 15;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (v2f){1.f + i, 2.f + tid * 1e-3f};
  for (int i = tid; i < 16 * kPlane + 512; i += 512) ex[i] = (v2f){1e-3f, 1e-3f};
  v2f* const wr = ex + t * kPlane + (tid >> 4);
  const v2f* const rd = ex + tid;
  const v4f* const rd4 = reinterpret_cast<const v4f*>(ex) + tid;
  float* const dst = sink + ((size_t)blockIdx.x * 512 + tid);
  const size_t row = (size_t)gridDim.x * 512;
  __syncthreads();
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == PK_FMA || MODE == MIX || MODE == MIXSTORE) {
#pragma unroll
      for (int rep = 0; rep < 12; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(a[(i + 1) & 15]));
    }
    if (MODE == PK_ADD) {
#pragma unroll
      for (int rep = 0; rep < 12; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
    }
    if (MODE == DPP_ADD) {               // plain f32 add with a cross-lane (quad_perm) source: 2 per pair
#pragma unroll
      for (int rep = 0; rep < 12; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[(i + 1) & 15].x));
          asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i].y) : "v"(a[(i + 1) & 15].y));
        }
    }
    if (MODE == DPP_XPOSE) {
      // The 16 x 16 transpose of k_synth7's exchange (thread t of a column holds 16 complex values,
      // wants element t of every other thread's) WITHOUT LDS: four butterfly stages over the 16
      // lanes of a DPP row, lane bit s against register-index bit s.  Per stage and register pair:
      // pick what goes (v_cndmask), move it to the partner lane t ^ (1 << s) (quad_perm for s = 0, 1;
      // row_shl:4 / row_shr:4 under complementary bank masks for s = 2; row_ror:8 for s = 3), put
      // what came into the right register (2 v_cndmask) -- per 32-bit half of the complex value.
#define XP_HALF(X, Y, MOVS)                                                                        \
      { float snd, rcv;                                                                             \
        asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(snd) : "v"(Y), "v"(X), "s"(hi));         \
        MOVS                                                                                        \
        asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(X) : "v"(rcv), "s"(hi));                 \
        asm volatile("v_cndmask_b32 %0, %1, %0, %2" : "+v"(Y) : "v"(rcv), "s"(hi)); }
#define XP_Q(PERM) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:" PERM " row_mask:0xf bank_mask:0xf" : "=v"(rcv) : "v"(snd));
#define XP_4 asm volatile("v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(rcv) : "v"(snd)); \
             asm volatile("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(rcv) : "v"(snd));
#define XP_8 asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(rcv) : "v"(snd));
#pragma unroll
      for (int sbit = 0; sbit < 4; ++sbit) {
        const unsigned long long hi = __builtin_amdgcn_ballot_w64((tid >> sbit) & 1);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (j & (1 << sbit)) continue;
          v2f& X = a[j];
          v2f& Y = a[j | (1 << sbit)];
          if (sbit == 0) { XP_HALF(X.x, Y.x, XP_Q("[1,0,3,2]")) XP_HALF(X.y, Y.y, XP_Q("[1,0,3,2]")) }
          else if (sbit == 1) { XP_HALF(X.x, Y.x, XP_Q("[2,3,0,1]")) XP_HALF(X.y, Y.y, XP_Q("[2,3,0,1]")) }
          else if (sbit == 2) { XP_HALF(X.x, Y.x, rcv = 0.f; XP_4) XP_HALF(X.y, Y.y, rcv = 0.f; XP_4) }
          else { XP_HALF(X.x, Y.x, XP_8) XP_HALF(X.y, Y.y, XP_8) }
        }
      }
    }
    if (MODE == ADD) {                   // plain f32 add, 2 per pair
#pragma unroll
      for (int rep = 0; rep < 12; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(a[(i + 1) & 15].x));
          asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].y) : "v"(a[(i + 1) & 15].y));
        }
    }
    if (MODE == SQRT) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
    }
    if (MODE == LDS_W64 || MODE == LDS_R64 || MODE == LDS_R128) {   // asm volatile: the compiler must not merge or drop them
      const unsigned wa = (unsigned)(reinterpret_cast<size_t>(wr) & 0xffff), ra = (unsigned)(reinterpret_cast<size_t>(rd) & 0xffff);
#pragma unroll
      for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (MODE == LDS_W64) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(wa), "v"(a[j]), "n"(j * 256) : "memory");
          else if (MODE == LDS_R64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[j]) : "v"(ra), "n"(j * 4104));
          else { v4f q; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"((unsigned)(tid * 16)), "n"((j & 7) * 8192)); a[j].x = q.x; a[j].y = q.w; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    if (MODE == EXCHANGE || MODE == MIX || MODE == MIXSTORE) {
      if (MODE != EXCHANGE) {
        const v4f* const g = reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(ex) + t * 20);
#pragma unroll
        for (int j = 0; j < 16; j += 4) { const v4f q = g[j >> 2]; a[j].x += q.x; a[j + 1].x += q.y; a[j + 2].x += q.z; a[j + 3].x += q.w; }
#pragma unroll
        for (int j = 0; j < 16; ++j) { const v2f q = ex[16 * kPlane + 16 * j + t]; a[j] += q; }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) wr[j * 32] = a[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = rd[j * kPlane];
      __syncthreads();
    }
    if (MODE == STORE || MODE == MIXSTORE) {
#pragma unroll
      for (int j = 1; j < 15; ++j) __builtin_nontemporal_store(a[j].x, dst + (size_t)((it * 14 + j) & 2047) * row);
    }
    if (MODE == STORE_DEF) {
#pragma unroll
      for (int j = 1; j < 15; ++j) dst[(size_t)((it * 14 + j) & 2047) * row] = a[j].x;
    }
    if (MODE == STORE4 || MODE == STORE4_DEF) {      // 16 bytes per lane: 1 KB contiguous per wave store, the same bytes per iteration
      v4f* const d4 = reinterpret_cast<v4f*>(sink) + ((size_t)blockIdx.x * 512 + tid);
#pragma unroll
      for (int j = 0; j < 14; j += 4) {
        const v4f q = {a[j].x, a[j].y, a[j + 1].x, a[j + 1].y};
        v4f* const at = d4 + (size_t)((it * 4 + (j >> 2)) & 511) * row;
        if (j < 12 || (it & 1)) { if (MODE == STORE4) __builtin_nontemporal_store(q, at); else *at = q; }
      }
    }
    if (MODE == STORE2) {                            // 8 bytes per lane
      v2f* const d2 = reinterpret_cast<v2f*>(sink) + ((size_t)blockIdx.x * 512 + tid);
#pragma unroll
      for (int j = 0; j < 7; ++j) __builtin_nontemporal_store(a[j], d2 + (size_t)((it * 7 + j) & 1023) * row);
    }
  }
  const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 7 && tid == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int MODE>
void drive(double seconds, float* out, float* sink, long long* clk, int grid) {
  const int iters = (MODE == STORE || (MODE >= STORE_DEF && MODE <= STORE2)) ? 200 : 2000;
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
  printf("%-9s %.2f s, %.3f us per iteration of a CU's 16 waves, in-kernel clock %.3f GHz", kNames[MODE], el, per_iter_us,
         (double)h[0] / ((double)h[1] * 10.0) );
  if (MODE == STORE || MODE == MIXSTORE || (MODE >= STORE_DEF && MODE <= STORE2)) printf(", %.2f TB/s", 14.0 * grid * 512 * 4 / per_iter_us / 1e6);
  printf("\n");
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "mix";
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  float *out, *sink; long long* clk;
  const int grid = 512;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, 64));
  CK(hipMalloc(&sink, (size_t)2048 * grid * 512 * 4));   // 2 GiB: 2048 rows of one float per thread
  if (!strcmp(mode, "idle")) { printf("idle\n"); fflush(stdout); std::this_thread::sleep_for(std::chrono::duration<double>(seconds)); return 0; }
#define CASE(M) if (!strcmp(mode, kNames[M])) { drive<M>(seconds, out, sink, clk, grid); return 0; }
  CASE(PK_FMA) CASE(PK_ADD) CASE(SQRT) CASE(LDS_W64) CASE(LDS_R64) CASE(LDS_R128) CASE(EXCHANGE) CASE(STORE) CASE(MIX) CASE(MIXSTORE) CASE(STORE_DEF) CASE(STORE4) CASE(STORE4_DEF) CASE(STORE2) CASE(DPP_ADD) CASE(ADD) CASE(DPP_XPOSE)
  printf("unknown mode %s\n", mode);
  return 1;
}
