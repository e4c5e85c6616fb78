// Round 6: what a CU mask on a stream does on this chip, and two reference curves under it.
//   (1) which CUs a masked stream's workgroups land on: distinct (XCC, SE, SH, CU) per XCC, so that "the mask's first n
//       bits are n / 8 CUs of every XCD" (api.cpp, option cu_count) is measured, not assumed;
//   (2) a store-only fill (16 bytes per lane, non-temporal) and (3) a packed-FMA loop with no memory traffic, at 256 /
//       224 / 192 / 160 / 128 / 96 / 64 CUs with package power and clocks beside each: what "bandwidth-bound" and
//       "issue-bound" look like under the mask on this box (the synthesis kernels' curves are in tools/bound_sweep.py).
// Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <glob.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_am_i(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // a little work so that the grid spreads instead of draining through the first CUs
  float v = (float)threadIdx.x;
  for (int i = 0; i < 2000; ++i) v = v * 1.0001f + 0.5f;
  if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffff00u) | ((xcc & 0xf) << 24) | (v == 1.f ? 1u : 0u);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) fill_nt(float* p, size_t n_vec, int per_wg) {
  // a workgroup writes per_wg runs of 4 KB, consecutive (the synthesis' 1 KB-per-wave stores, row-linear)
  size_t base = (size_t)blockIdx.x * per_wg * 256;
  const v4u q = {1u, 2u, 3u, 4u};
  for (int i = 0; i < per_wg; ++i) {
    const size_t k = base + (size_t)i * 256 + threadIdx.x;
    if (k < n_vec) __builtin_nontemporal_store(q, reinterpret_cast<v4u*>(p) + k);
  }
}

__global__ void __launch_bounds__(256) fma_loop(float* out, int iters) {
  v2f a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f};
  const v2f c = {1.0001f, 0.9999f}, d = {(float)threadIdx.x, 1.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(d));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(d));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(d));
    }
  }
  if (a0.x + a1.x + a2.x + a3.x == 12345.f) out[0] = a0.y;
}

// package power / clocks of the card under load: the one of the box's cards that draws the most at the moment of the read
struct Sysfs {
  std::vector<std::string> hw, pf;
  Sysfs() {
    glob_t g;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*", 0, nullptr, &g) == 0) {
      for (size_t i = 0; i < g.gl_pathc; ++i)
        for (const char* name : {"/power1_average", "/power1_input"}) {
          const std::string f = std::string(g.gl_pathv[i]) + name;
          if (FILE* fp = fopen(f.c_str(), "r")) { fclose(fp); hw.push_back(g.gl_pathv[i]); pf.push_back(f); break; }
        }
    }
    globfree(&g);
    // the card this process drives, by PCI address; else the busiest card at each read
    char id[64] = {};
    if (hipDeviceGetPCIBusId(id, sizeof id, 0) == hipSuccess) {
      for (char* c = id; *c; ++c) *c = (char)tolower(*c);
      const std::string pat = std::string("/sys/bus/pci/devices/") + id + "/hwmon/hwmon*";
      if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc > 0)
        for (const char* name : {"/power1_average", "/power1_input"}) {
          const std::string f = std::string(g.gl_pathv[0]) + name;
          if (FILE* fp = fopen(f.c_str(), "r")) { fclose(fp); hw.assign(1, g.gl_pathv[0]); pf.assign(1, f); break; }
        }
      globfree(&g);
    }
  }
  static double num(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    double v = -1;
    if (f) { if (fscanf(f, "%lf", &v) != 1) v = -1; fclose(f); }
    return v;
  }
  // the starred line of a pp_dpm_* table, MHz
  static double dpm(const std::string& dev, const char* name) {
    FILE* f = fopen((dev + "/" + name).c_str(), "r");
    if (!f) return -1;
    char line[128];
    double v = -1;
    while (fgets(line, sizeof line, f))
      if (strchr(line, '*')) { const char* c = strchr(line, ':'); if (c) v = atof(c + 1); }
    fclose(f);
    return v;
  }
  void read(double* w, double* sclk, double* mclk, double* fclk) const {
    *w = *sclk = *mclk = *fclk = -1;
    for (size_t i = 0; i < hw.size(); ++i) {
      const double p = num(pf[i]) / 1e6;
      if (p > *w) {
        const std::string dev = hw[i].substr(0, hw[i].find("/hwmon"));   // (.../device/hwmon/hwmonN or the PCI device's)
        *w = p; *sclk = num(hw[i] + "/freq1_input") / 1e9; *mclk = dpm(dev, "pp_dpm_mclk"); *fclk = dpm(dev, "pp_dpm_fclk");
      }
    }
  }
};

int main() {
  Sysfs sy;
  printf("sysfs: %zu cards with a power reading\n", sy.hw.size());
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  const int counts[] = {256, 224, 192, 160, 128, 96, 64};
  unsigned* d_where;
  const int n_wg = 16384;
  CK(hipMalloc(&d_where, n_wg * 4));
  const size_t bytes = (size_t)16 << 30;
  float* buf;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("| CUs in mask | distinct CUs seen | per XCC | fill ms | fill TB/s | fill W | fill sclk | mclk | fclk | fma ms | fma W | fma sclk |\n");
  printf("|---|---|---|---|---|---|---|---|---|---|---|---|\n");
  for (int n_cu : counts) {
    uint32_t mask[16] = {};
    for (int i = 0; i < n_cu; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, 16, mask));
    hipLaunchKernelGGL(where_am_i, dim3(n_wg), dim3(64), 0, st, d_where);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(n_wg);
    CK(hipMemcpy(h.data(), d_where, n_wg * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> all;
    std::vector<std::set<unsigned>> per(16);
    for (unsigned v : h) { all.insert(v & ~1u); per[(v >> 24) & 0xf].insert(v & ~1u); }
    std::string perx;
    for (int x = 0; x < 16; ++x) if (!per[x].empty()) perx += std::to_string(per[x].size()) + " ";
    auto run_for = [&](double seconds, auto&& launch, double* ms_out, double* w_out, double* sclk_out, double* mclk, double* fclk) {
      // warm
      for (int i = 0; i < 3; ++i) launch();
      CK(hipStreamSynchronize(st));
      const auto t0 = std::chrono::steady_clock::now();
      double sum_ms = 0, sw = 0, sc = 0, sm = 0, sf = 0;
      int n = 0, ns = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 8; ++i) launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.4 * seconds) {
          sum_ms += ms / 8; ++n;
          double w, c, m, f;
          sy.read(&w, &c, &m, &f);
          sw += w; sc += c; sm += m; sf += f; ++ns;
        }
      }
      *ms_out = sum_ms / (n ? n : 1); *w_out = sw / (ns ? ns : 1); *sclk_out = sc / (ns ? ns : 1);
      if (mclk) *mclk = sm / (ns ? ns : 1);
      if (fclk) *fclk = sf / (ns ? ns : 1);
    };
    double f_ms, f_w, f_ck, mclk, fclk, a_ms, a_w, a_ck;
    const size_t n_vec = bytes / 16;
    const int per_wg = 64;
    const unsigned wgs = (unsigned)((n_vec + (size_t)per_wg * 256 - 1) / ((size_t)per_wg * 256));
    run_for(2.0, [&] { hipLaunchKernelGGL(fill_nt, dim3(wgs), dim3(256), 0, st, buf, n_vec, per_wg); }, &f_ms, &f_w, &f_ck, &mclk, &fclk);
    run_for(2.0, [&] { hipLaunchKernelGGL(fma_loop, dim3(256 * 32), dim3(256), 0, st, buf, 400); }, &a_ms, &a_w, &a_ck, nullptr, nullptr);
    printf("| %d | %zu | %s| %.3f | %.2f | %.0f | %.2f | %.0f | %.0f | %.3f | %.0f | %.2f |\n", n_cu, all.size(), perx.c_str(), f_ms,
           bytes / f_ms / 1e9, f_w, f_ck, mclk, fclk, a_ms, a_w, a_ck);
    fflush(stdout);
    CK(hipStreamDestroy(st));
  }
  return 0;
}
