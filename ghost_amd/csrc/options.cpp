// options.cpp -- see options.h
#include "options.h"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "kernels.h"

namespace gcwt {
namespace {

// layout / kernel selection: every choice computes the same rows (tests compare them)
const char* const kSelect[] = {"synth16", "synth_cols", "fuse_blocks", "slow_fft", "level_streams", "interp_grid",
                               "synth_streams", "interp_lgnb", "merge_levels", "split_levels", "interp",
                               "batch_bytes", "stage_floats", "fullband_group", "blockconv",
                               "direct_max_len", "graphs", "plan_threads", "fullband_cache_mb", "auto_threshold_ppb", "auto_kappa_ppb", "auto_oob_ppt", "synth7_order", "host_widen", "host_threads", "fullband4", "cu_count", "synth7_narrow_r", "fold_mean"};
// accuracy-changing or measurement hooks: libghostcwt_measure.so only
const char* const kMeasureOnly[] = {"halo_margin", "interp_q", "interp_taps", "interp_min_r", "prune_inputs", "clock_phases",
                                    "synth_kernel", "synth_drop_stores", "clock_probe", "synthi_pad_kb",
                                    // measured-slower kernels: k_synthp (synthp.hip; ties with k_synthi at a lower clock: profiles/r05_synth_study.md)
                                    "synthp", "synthp_lgnb", "synthp_help"};
// the two budgets the product library also takes from the environment
const char* const kEnvBudgets[] = {"batch_bytes", "stage_floats"};

std::mutex g_mu;
std::map<std::string, long long>& table() {
  static std::map<std::string, long long> t;
  return t;
}

template <size_t N>
bool in_list(const char* name, const char* const (&list)[N]) {
  for (const char* s : list)
    if (!strcmp(s, name)) return true;
  return false;
}

bool from_env(const char* name, long long* v) {
  std::string var = "GHOSTCWT_";
  for (const char* c = name; *c; ++c) var.push_back((char)toupper((unsigned char)*c));
  const char* e = getenv(var.c_str());
  if (!e) return false;
  *v = *e ? atoll(e) : 1;
  return true;
}

bool lookup(const char* name, long long* v) {
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = table().find(name);
    if (it != table().end()) { *v = it->second; return true; }
  }
  if (kMeasureBuild || in_list(name, kEnvBudgets)) return from_env(name, v);
  return false;
}

}  // namespace

long long option_or(const char* name, long long dflt) {
  long long v;
  return lookup(name, &v) ? v : dflt;
}

bool option_is_set(const char* name) {
  long long v;
  return lookup(name, &v);
}

int option_set(const char* name, long long value, bool clear) {
  if (!name) return -1;
  const bool measure_only = in_list(name, kMeasureOnly);
  if (!measure_only && !in_list(name, kSelect)) return -1;
  if (measure_only && !kMeasureBuild) return -2;
  std::lock_guard<std::mutex> lock(g_mu);
  if (clear) table().erase(name);
  else table()[name] = value;
  return 0;
}

}  // namespace gcwt
