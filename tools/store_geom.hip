// Store-only probe with k_synthi's geometry (round 4, VERDICT r03 task 2): what stops its stores at
// 5.0 - 5.4 TB/s when a linear fill reaches 6.8 - 7.0 on the same box?
//   hipcc -O3 --offload-arch=gfx950 tools/store_geom.hip -o /tmp/sg && /tmp/sg
// Model of the kernel's store side: a workgroup of 4 waves owns a column range of `visit` samples of
// one channel and walks rows (scales) `ns` at a time for `run` passes; inside a (row, range) visit the
// waves write 1 KB runs round-robin (16-byte stores per lane), exactly as pass B does.  3 workgroups
// per CU (42 KB of LDS each), `nt` stores by default.  Swept: bytes per visit, order of the work items
// over the grid (block-major, pass-major, channels fastest), row pitch, store policy, waves in flight.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Geom {
  int n_ch, n_rows;          // channels, rows (scales of the level) per channel
  long long pitch;           // samples between rows
  int n_samples;             // samples per row that are written
  int visit;                 // samples of one (row, range) visit: hop * R * nb in the kernel
  int ns, run;               // rows per pass, passes per workgroup
  int order;                 // 0: items x-fastest, block outer / pass inner (the kernel's list order)
                             // 1: items x-fastest, pass outer / block inner
                             // 2: channels fastest (grid (C, items)), block outer
                             // 3: channels fastest, pass outer
  int policy;                // 0 default, 1 nt
  int rows_total;            // row stride of a channel in rows (S of the plan: 100)
};

template <int POLICY>
__global__ void __launch_bounds__(256, 3) k_store(float* __restrict__ out, const Geom g, int n_blk, int n_runs) {
  extern __shared__ char lds[];
  (void)lds;
  int item, ch;
  if (g.order >= 2) { ch = blockIdx.x; item = blockIdx.y; } else { item = blockIdx.x; ch = blockIdx.y; }
  int blk, rn;
  if (g.order & 1) { rn = item / n_blk; blk = item - rn * n_blk; } else { blk = item / n_runs; rn = item - blk * n_runs; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long col0 = (long long)blk * g.visit;
  const int len = min(g.visit, g.n_samples - (int)col0);
  if (len <= 0) return;
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 val = {1.f, 2.f, 3.f, (float)item};
  for (int ps = 0; ps < g.run; ++ps) {
    for (int s = 0; s < g.ns; ++s) {
      const int row = (rn * g.run + ps) * g.ns + s;
      if (row >= g.n_rows) continue;
      float* dst = out + ((long long)ch * g.rows_total + row) * g.pitch + col0;
      for (int wt = wave; wt * 256 < len; wt += 4) {
        const int smp = wt * 256 + 4 * lane;
        if (smp + 4 <= len) {
          if (POLICY == 1) __builtin_nontemporal_store(val, reinterpret_cast<v4*>(dst + smp));
          else *reinterpret_cast<v4*>(dst + smp) = val;
        }
      }
    }
  }
}

typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ void k_fill(vf4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  const vf4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n; i += st) __builtin_nontemporal_store(v, p + i);
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)60 << 30;
  float* a; CK(hipMalloc(&a, bytes));
  CK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k_store<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 42 * 1024));
  CK(hipFuncSetAttribute((const void*)k_store<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 42 * 1024));
  auto run = [&](const char* name, Geom g, int lds_kb = 42) {
    const int n_blk = (g.n_samples + g.visit - 1) / g.visit;
    const int n_pass = (g.n_rows + g.ns - 1) / g.ns;
    const int n_runs = (n_pass + g.run - 1) / g.run;
    const int items = n_blk * n_runs;
    dim3 grid = g.order >= 2 ? dim3(g.n_ch, items) : dim3(items, g.n_ch);
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
      CK(hipEventRecord(e0));
      if (g.policy) hipLaunchKernelGGL(k_store<1>, grid, dim3(256), lds_kb * 1024, 0, a, g, n_blk, n_runs);
      else hipLaunchKernelGGL(k_store<0>, grid, dim3(256), lds_kb * 1024, 0, a, g, n_blk, n_runs);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double gb = (double)g.n_ch * g.n_rows * (g.n_samples / 4 * 4) * 4.0 / 1e9;
    printf("%-86s items %6d  %7.3f ms (best %7.3f)  %6.2f TB/s\n", name, items, ms[ms.size() / 2], ms[0],
           gb / ms[ms.size() / 2]); fflush(stdout);
  };
  {
    std::vector<float> ms;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (vf4*)a, ((size_t)51 << 30) / 16);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("linear nt fill of 51 GiB: %.3f ms  %.2f TB/s\n", ms[ms.size() / 2], 51.0 * 1.073741824 / ms[ms.size() / 2]);
  }
  const long long pitch = 1000032;       // the bench's row pitch (rows start on 128-byte lines)
  char name[256];
  // the headline's interpolated levels: R = 32 (nb 2: 54 KB visits, ns 4), 64 (54 KB, ns 8), 128 (105 KB, ns 8)
  struct Lv { int R, hop, nb, ns, rows; } lvs[] = {{16, 212, 2, 4, 15}, {32, 212, 2, 4, 15}, {64, 212, 1, 8, 15}, {128, 206, 1, 8, 18}};
  for (const Lv& lv : lvs) {
    for (int order = 0; order < 4; ++order) {
      Geom g{128, lv.rows, pitch, 1000000, lv.hop * lv.R * lv.nb, lv.ns, (lv.rows + lv.ns - 1) / lv.ns, order, 1, 100};
      snprintf(name, sizeof name, "R %3d visit %6.1f KB ns %d run all  order %d nt", lv.R, g.visit * 4 / 1024.0, lv.ns, order);
      run(name, g);
    }
  }
  // bytes per visit (R = 32-like rows: 15 rows, ns 4), order 0
  for (int visit : {3392, 6784, 13568, 27136, 54272, 108544, 250000}) {
    Geom g{128, 15, pitch, 1000000, visit, 4, 4, 0, 1, 100};
    snprintf(name, sizeof name, "visit sweep: %7.1f KB per (row, range), ns 4, run all, order 0 nt", visit * 4 / 1024.0);
    run(name, g);
  }
  // ns sweep (rows a workgroup alternates between inside a pass) at 54 KB visits
  for (int ns : {1, 2, 4, 8, 15}) {
    Geom g{128, 15, pitch, 1000000, 13568, ns, (15 + ns - 1) / ns, 0, 1, 100};
    snprintf(name, sizeof name, "ns sweep: %2d rows per pass, 53 KB visits, run all, order 0 nt", ns);
    run(name, g);
  }
  // run length (passes per workgroup): short runs = more, shorter-lived workgroups
  for (int rn : {1, 2, 4}) {
    Geom g{128, 15, pitch, 1000000, 13568, 4, rn, 0, 1, 100};
    snprintf(name, sizeof name, "run sweep: %d passes per workgroup, 53 KB visits, ns 4, order 0 nt", rn);
    run(name, g);
  }
  // policy
  for (int pol = 0; pol < 2; ++pol) {
    Geom g{128, 15, pitch, 1000000, 13568, 4, 4, 0, pol, 100};
    snprintf(name, sizeof name, "policy %s, 53 KB visits, ns 4", pol ? "nt" : "default");
    run(name, g);
  }
  // row pitch: partition camping?  (bytes between rows)
  for (long long dp : {-32LL, 0LL, 32LL, 64LL, 96LL, 224LL, 480LL, 992LL, 2016LL, 4064LL, 16352LL, 48544LL, 65504LL}) {
    Geom g{128, 15, pitch + dp, 1000000, 13568, 4, 4, 0, 1, 100};
    snprintf(name, sizeof name, "pitch %9lld B (%+6lld samples), 53 KB visits, ns 4, order 0 nt", (pitch + dp) * 4, dp);
    run(name, g);
  }
  // occupancy: workgroups per CU through the LDS size
  for (int kb : {20, 42, 64, 100}) {
    Geom g{128, 15, pitch, 1000000, 13568, 4, 4, 0, 1, 100};
    snprintf(name, sizeof name, "occupancy: %3d KB LDS per workgroup, 53 KB visits, ns 4, order 0 nt", kb);
    if (kb > 42) { CK(hipFuncSetAttribute((const void*)k_store<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024)); }
    run(name, g, kb);
  }
  return 0;
}
