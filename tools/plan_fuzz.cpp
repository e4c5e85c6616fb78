// Sanitizer harness for the host-side planner (planner.cpp has no HIP call): thousands of random
// plan requests -- layouts, sampling rates, wavelets, epochs, forced time blocks -- under
// AddressSanitizer and UBSan on the CPU (GPU sanitizers are not available on this pool).
//   cd ghost_amd/csrc && g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer \
//     -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I. -I../../include ../../tools/plan_fuzz.cpp planner.cpp options.cpp \
//     -o /tmp/plan_fuzz && /tmp/plan_fuzz        (-fsanitize=thread with PLAN_FUZZ_N=400 for the level designs' threads; 6 000 requests incl. long mode, all four precisions, many-epoch layouts and the block-convolution invariants, 2 min; round 4: clean)
#include "planner.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <algorithm>
#include <cmath>
int main() {
  std::mt19937_64 rng(7);
  int ok = 0, refused = 0, bc_plans = 0;
  const int n_it = getenv("PLAN_FUZZ_N") ? atoi(getenv("PLAN_FUZZ_N")) : 6000;
  for (int it = 0; it < n_it; ++it) {
    gcwt_params prm{};
    const double fss[] = {200.0, 1000.0, 1250.0, 30000.0};
    prm.fs = fss[rng() % 4];
    prm.n_channels = 1 + (int)(rng() % 4);
    const int64_t ns[] = {17, 500, 4096, 4097, 10000, 33333, 70000, 150000, 1000000, 5000000, 18000000};
    prm.n_samples = ns[rng() % 11];
    { const double gs[] = {1, 2, 3, 4, 6}; prm.gamma = gs[rng() % 5]; }
    prm.beta = 1.5 + (double)(rng() % 800) / 10.0;
    if (rng() % 2) { prm.gamma = 3; prm.beta = 20; }
    std::vector<double> f;
    const int nf = 1 + (int)(rng() % 40);
    const double lo = std::max(1e-4 * prm.fs, 10.0 * prm.fs / std::max<int64_t>(prm.n_samples, 20)), hi = 0.47 * prm.fs;
    for (int i = 0; i < nf; ++i) f.push_back(lo < hi ? lo * std::pow(hi / lo, (double)(rng() % 1000) / 999.0) : 0.4 * prm.fs);
    std::sort(f.rbegin(), f.rend());
    prm.n_freqs = nf; prm.freqs_hz = f.data();
    std::vector<int64_t> eb;
    int64_t cur = 0;
    const int ne = rng() % 6 == 0 ? 200 : 1 + (int)(rng() % 5);      // now and then: many short epochs
    for (int e = 0; e < ne && cur + 8 < prm.n_samples; ++e) {
      const int64_t a = cur + (int64_t)(rng() % 3), b = e == ne - 1 ? prm.n_samples : std::min<int64_t>(prm.n_samples, a + 4 + (int64_t)(rng() % (prm.n_samples / ne + 1)));
      if (b - a > 3) { eb.push_back(a); eb.push_back(b); }
      cur = b;
    }
    if (eb.empty()) { eb = {0, prm.n_samples}; }
    prm.n_epochs = (int)eb.size() / 2; prm.epoch_bounds = eb.data();
    prm.out_mode = (int)(rng() % 3);
    const int mf[] = {0, 0, 12, 13, 14, 16, 21, 23, 24};          // 23 / 24: long mode (round 4)
    prm.max_fft_log2 = mf[rng() % 9];
    prm.precision = (int)(rng() % 4);                              // default / fast / high / exact: low cut, ramps, full support, no decimated path
    if (rng() % 8 == 0) prm.support_tol = 1e-7;
    gcwt::HostPlan hp;
    std::string err;
    const int rc = gcwt::build_host_plan(prm, &hp, &err);
    if (rc == 0) ++ok; else ++refused;
    if (rc == 0) {
      // block convolution (round 4): every such scale in exactly one group, whole inside the group's window
      std::vector<int> seen(hp.scales.size(), 0);
      int in_groups = 0;
      for (const auto& g : hp.bc_groups) {
        if (g.hop < 64 || g.hop % 64 || g.back % 64 || g.count < 1 || g.first != in_groups) { printf("bad group\n"); return 1; }
        for (int k = g.first; k < g.first + g.count; ++k) {
          const gcwt::ScalePlan& sp = hp.scales[hp.bc_order[k]];
          const int64_t ahead = (sp.length - 1) / 2, behind = sp.length - 1 - ahead;
          if (sp.method != GCWT_SCALE_BLOCKCONV || sp.blockconv_index != k || behind > g.back ||
              g.back + g.hop + ahead > 4096 || seen[hp.bc_order[k]]++) { printf("bad member\n"); return 1; }
        }
        in_groups += g.count;
      }
      int n_bc = 0;
      for (const auto& sp : hp.scales) n_bc += sp.method == GCWT_SCALE_BLOCKCONV;
      if (n_bc != in_groups || n_bc != hp.n_blockconv || (int)hp.bc_order.size() != n_bc ||
          (n_bc > 0 && hp.bc_chunk_blocks < 1) || !(hp.bc_fill > 0.0 && hp.bc_fill <= 1.0)) { printf("bad count\n"); return 1; }
      for (const auto& sp : hp.scales)
        if (sp.method == GCWT_SCALE_DIRECT && sp.length > 256) { printf("direct too long\n"); return 1; }
      bc_plans += n_bc > 0;
    }
  }
  printf("plans ok %d refused %d, with block-convolution scales %d\n", ok, refused, bc_plans);
  return 0;
}
