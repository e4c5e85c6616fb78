// api.cpp -- C ABI of libghostcwt.so (include/ghostcwt.h): plan objects, device
// workspace, and the orchestration of one transform.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/ghostcwt.h"
#include "../../include/ghostcwt_debug.h"
#include <memory>

#include "host_out.h"
#include "interp.h"
#include "kernels.h"
#include "morse_exact.h"
#include "options.h"
#include "planner.h"

using namespace gcwt;

namespace {

thread_local std::string g_err;


int set_err(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

int hip_err(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? GCWT_ERR_NOMEM : GCWT_ERR_HIP;
}

// Nothing may unwind across the C ABI: entry points that allocate run inside this.
template <typename F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return set_err(GCWT_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return set_err(GCWT_ERR_INVALID, std::string("internal error: ") + e.what());
  } catch (...) {
    return set_err(GCWT_ERR_INVALID, "internal error");
  }
}

#define HIP_TRY(call)                                  \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return hip_err(e_, #call);   \
  } while (0)

struct EpochDev {
  SynthItemDev* items = nullptr;     // 16-column kernel: levels the production kernel does not take
  int n_items = 0;
  SynthLevelDev* levels = nullptr;
  Synth7Item* items7 = nullptr;      // production kernel
  Synth7Level* levels7 = nullptr;
  int n_items7 = 0;
  Synth7Item* items7w = nullptr;     // ... its wide-halo instantiation (shifted-band levels with halo > 48)
  int n_items7w = 0;
  Synth7Item* items7n = nullptr;     // ... its 16-column instantiation (R = 2: gcwt_plan::synth7_narrow_r)
  int n_items7n = 0;
  SynthiItem* items_i = nullptr;     // interpolating kernel (synthi.hip)
  SynthiLevel* levels_i = nullptr;
  int n_items_i = 0;
  SynthpItem* items_p[2] = {nullptr, nullptr};   // pipelined interpolating kernel (synthp.hip); [1]: levels with I = 4
  SynthpLevel* levels_p = nullptr;
  int n_items_p[2] = {0, 0};
  PredLevel* pred_levels = nullptr;              // precision = auto / high: what each level's x_R holds (detect.hip)
};

// full-band responses kept on the device across executes (one P-point row per (scale, FFT length)); counted in
// gcwt_plan_info.workspace_bytes
constexpr int64_t kFullbandCacheBytes = (int64_t)16 << 30;   // 193 responses of 2^22 bins (config 5, precision = exact) are 6.2 GB

enum Stage { ST_MEAN = 0, ST_FWD, ST_DECIM, ST_BLOCK, ST_SYNTH, ST_DIRECT, ST_FULLBAND, ST_INTERP, ST_BLOCKCONV, ST_COUNT };

}  // namespace

struct gcwt_plan {
  HostPlan hp;
  bool uploaded = false;
  int profiling = 0;          // gcwt_plan_set_profiling: 0 off, 1 every stage between events, 2 the synthesis kernels only
  int synth_cols = 32;        // columns per workgroup of k_synth7 (GHOSTCWT_SYNTH_COLS=16|32)
  int synth7_narrow_r = 2;    // option synth7_narrow_r: levels of decimation <= this take the 16-column instantiation whatever
                              // synth_cols says (0: none).  R = 2 has seven scales on the headline grid, so a workgroup's prologue
                              // (its blocks' samples, their transforms) is a quarter of its life; 256-thread workgroups are three
                              // to a CU instead of two and hide it better: 1.17 against 1.26 ms for the level, while R = 4 and 8
                              // (fifteen scales) lose 8 % that way (profiles/r06_bound.md 5)
  bool fuse_blocks = true;    // k_synth7 makes its own block spectra from x_R (GHOSTCWT_FUSE_BLOCKS=0: separate pass)
  int synth_kernel = 7;       // 7: k_synth7; 8 (measure build only): producer/consumer waves (synth8.hip,
                              // measured slower: DESIGN.md 5) -- GHOSTCWT_SYNTH_KERNEL
  // Every GHOSTCWT_* knob is read ONCE, when the plan is created; an execute never looks at the
  // environment.
  bool prune_inputs = true;   // first-pass inputs above a scale's band skipped (kernels.hip: k_scale_windows);
                              // GHOSTCWT_PRUNE_INPUTS=0 computes them all (A/B runs)
  bool fast_fft = true;       // GHOSTCWT_SLOW_FFT=1: the generic radix-2 passes (A/B runs, tests)
  int drop_stores = 0;        // measure build only: GHOSTCWT_SYNTH_DROP_STORES (kernels.h)
  bool clock_probe = false;   // measure build only: GHOSTCWT_CLOCK_PROBE
  bool use_synth16 = false;   // GHOSTCWT_SYNTH16=1: 16-column kernel for every output mode (A/B tests)
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t aux[2] = {nullptr, nullptr};   // the level passes of a batch run beside each other (run_pipeline)
  bool level_streams = true;  // GHOSTCWT_LEVEL_STREAMS=0: everything on `stream`
  bool use_synthp = false;    // option synthp = 1: q = 2 levels with I <= 256 go to the pipelined kernel (synthp.hip); it
                              // ties with k_synthi on the headline and loses at R = 8 (profiles/r05_synth_study.md): off
  bool use_graphs = true;     // option graphs: small device-resident executes are replayed as a HIP graph
  bool fullband4 = true;      // option fullband4 = 0: 16 384-point segments by the two-pass kernels (A/B, tests)
  int fullband_group = 0;     // option fullband_group: rows per group of the fused full-band row pass (0: the kernel's default)
  int64_t fullband_cache_cap = 0;   // bytes of full-band responses kept across executes (set at upload: option
                                    // fullband_cache_mb, else a quarter of the memory free then, at most 16 GiB)
  bool hfull_cache_full = false;    // a response could not be allocated: the cache stays as it is
  int synth7_order = 0;       // option synth7_order: order of the k_synth7 work items (A/B runs)
  int synthp_help = -1;       // option synthp_help: share (of 128) of a round's tasks the producer waves take (A/B runs; default: by level)
  int synthp_lgnb = -1;       // option synthp_lgnb: blocks per k_synthp workgroup forced (A/B runs)
  int interp_lgnb = -1;       // option interp_lgnb: blocks per k_synthi workgroup forced (A/B runs)
  int interp_grid = -1;       // GHOSTCWT_INTERP_GRID=0|1: k_synthi's grid order forced (default: by the number of channel slots)
  bool last_fold = false;     // the last run took the channel sums inside the forward column pass
  bool fold_mean = true;      // option fold_mean = 0: the channel sums by a pass of their own over x even where the forward column pass could take them (run_pipeline)
  bool synth_streams = false; // GHOSTCWT_SYNTH_STREAMS=1: the interpolating kernel runs beside k_synth7 on aux[0] (its store-bound
                              // workgroups share the CUs with the arithmetic-bound ones: measured equal on the headline,
                              // profiles/r03_synth_study.md); default: one after the other, so that per-kernel times add up
  hipStream_t cur = nullptr;  // the stream the stage in hand is launched on (profiling spans follow it)
  // workspace
  bool high_precision = true; // gcwt_params.precision: float64 forward transform (fwd64.hip) + per-level low cut
  double2* d_y = nullptr;     // [slots][rows][4096] float64 intermediate of the forward transform
  int64_t y_stride = 0;
  double2* d_tw64 = nullptr;  // float64 twiddle tables (fwd64_fill_tables)
  float2* d_x = nullptr;      // [C][max_p]   spectrum (k1-major)
  float2* d_xr = nullptr;     // [C][max_xr]  decimated analytic signals, all levels
  float2* d_xb = nullptr;     // [C][max_xb]  block spectra, all levels
  float2* d_bank = nullptr;   // [S][B]
  float* d_gain = nullptr;    // [S][B] |H|
  float* d_gain_lv = nullptr; // the same rows in level-list order, transposed for k_synth7 (+8 zero rows)
  float2* d_half_tw = nullptr; // [levels][256] exp(-i pi k/(256 R))
  float2* d_psi = nullptr;    // direct kernels: running sums of the taps (kernels.hip: k_build_direct)
  float2* d_psi_tail = nullptr; // [n_direct] sum of all taps of each
  float2* d_psi_lit = nullptr;  // the literal taps, d_psi's layout (gcwt_direct_kernel)
  unsigned long long* d_probe = nullptr;   // GHOSTCWT_CLOCK_PROBE=1: [cycles, 100 MHz ticks] of the synthesis workgroups
  double* d_amps = nullptr;   // kept spectrum samples A_j of every scale (planner.h: amps)
  float2* d_xs = nullptr;     // [C][xs_stride] shifted slice of the spectrum of the level in hand (levels with a
  int64_t xs_stride = 0;      //   band shift: heavy-tailed wavelets); xs_stride = the largest such level's M
  float2* d_z = nullptr;      // [z_sets][slots][max_p]  full-band scales, up to four at a time: row-transformed spectrum * response
  int64_t z_half = 0;         //              elements between two of the set
  int z_sets = 1;
  float2* d_hfull = nullptr;  // [z_sets][max_p]  full-band responses of the scales in hand (when the cache below is full)
  // Full-band responses are an O(n_bins P) fp64 evaluation each: computed once per (scale, FFT
  // length) and kept on the device while they fit 16 GiB, reused by every later batch and execute.
  std::map<std::pair<int, int>, float2*> hfull_cache;
  int64_t hfull_cache_bytes = 0;
  float2* d_tw4096 = nullptr; // exp(-2 pi i j/4096), j < 2048
  float2* d_tw256 = nullptr;  // exp(+2 pi i q/256)
  float2* d_level_tw = nullptr;
  double* d_sums = nullptr;   // [C] sums (+ the partial sums and counters of k_channel_sum)
  int32_t* d_scale_list = nullptr;
  int n_listed = 0;           // entries of d_scale_list (all levels)
  int32_t* d_scale_aux = nullptr;   // per list entry: demodulation bin | (even kernel length) << 16 (synthi.hip)
  float* d_interp_coef = nullptr;   // interpolator coefficients of the interpolated levels
  BankScale* d_bank_sc = nullptr;
  DirectScale* d_direct_sc = nullptr;
  float2* d_bc_h = nullptr;       // [n_blockconv][4096]  block convolution: responses, in HostPlan::bc_order
  int32_t* d_bc_rows = nullptr;   // [n_blockconv]        their output rows
  float2* d_bc_x = nullptr;       // [bc_chunk_blocks][C][4096]  spectra of the blocks in hand
  float2* d_bc_tw = nullptr;      // [4096]               the middle twiddles in k_bc_scales' order
  std::vector<EpochDev> ep_dev;
  int64_t max_direct_len = 0;
  // staging for host-side callers
  float* d_in = nullptr;
  size_t d_in_bytes = 0;
  void* d_out = nullptr;
  size_t d_out_bytes = 0;
  HostOut host_out;          // pinned staging ring for host results
  // profiling
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Span { int stage; hipEvent_t a, b; };
  std::vector<Span> spans;
  gcwt_timings last{};
  bool have_timings = false;
  bool have_means = false;
  // precision = auto / high: the detector (detect.hip).  d_hist: band energies of the spectrum per slot, filled by the
  // forward row pass; d_pred: per scale, the largest predicted loss over the slots of an execute.
  bool detect = false;               // the plan predicts (float64 forward transform, no long mode, some spectral scale)
  float* d_hist = nullptr;
  float* d_bands = nullptr;          // [slots][kSpecBands]: the rows of d_hist added up (detect.hip: k_band_sums)
  float* d_pred = nullptr;
  hipStream_t det_stream = nullptr;  // the predictions are made beside the level passes and the synthesis, not behind them
  int32_t* d_scale_level = nullptr;
  int32_t* d_scale_length = nullptr;   // the reference kernel's L per scale (capped at 2^30)
  std::vector<float> last_pred;      // the last execute's predictions (host)
  float* h_pred = nullptr;           // ... as they arrive: page-locked, so that the 4 S bytes do not go through a staging copy
  float last_worst = 0.f;
  int last_rerouted = 0;
  float auto_threshold = 1.5e-6f;    // option auto_threshold_ppb.  The prediction is an r.m.s. figure calibrated on noise-like rows
                                     // (1.6e-7 D); a row that is nearly a sinusoid (narrow wavelets, gamma = 6) has a crest
                                     // factor three times smaller, so its gate metric reads three times the prediction: the
                                     // threshold leaves a factor 6.7 to the 1e-5 gate (benign recordings predict 2 - 4e-7)
  float kappa_eps = 1.6e-7f;         // predicted loss = kappa_eps sqrt(E_level W_s / E_s); option auto_kappa_ppb
  float oob_tol = 2.5e-8f;            // ... or oob_tol sqrt(E_out / E_s), what the level leaves out; option auto_oob_ppt (1e-12)
  int last_batch_slots = 0;          // slots of the last batch that ran (gcwt_debug_precision_terms)
  int last_batch = 0, last_rows = 0;
  double last_pt = 0;
  // scales made again by the exact paths: a sub-plan with precision = exact over ALL the scales, for every channel [0] or
  // for one [1], made on first need; a run of it is masked to the scales wanted and writes straight into this plan's rows
  gcwt_plan* sub_plan[2] = {nullptr, nullptr};
  bool is_sub_plan = false;          // this plan IS such a sub-plan (its response cache takes half the share)
  // a masked run (this plan IS such a sub-plan): run_mask[scale] != 0 -> make the row; nothing else is touched
  const unsigned char* run_mask = nullptr;
  unsigned char* d_run_mask = nullptr;
  // Small device-resident executes are launch-bound (config 1: fifteen kernels of a few microseconds each): the
  // second execute with the same arguments is captured into a graph, later ones replay it
  struct GraphKey {
    const void* x = nullptr; const void* out = nullptr;
    int64_t r0 = 0, r1 = 0, row_len = 0; bool reuse = false;
    bool operator==(const GraphKey& o) const {
      return x == o.x && out == o.out && r0 == o.r0 && r1 == o.r1 && row_len == o.row_len && reuse == o.reuse;
    }
  };
  GraphKey graph_key, graph_seen;
  hipGraphExec_t graph_exec = nullptr;
  bool graph_seen_valid = false, graph_failed = false;
  int64_t row_pitch = 0;     // device output rows, samples; 0 = dense
};

namespace {

template <typename T>
int dev_alloc(T** p, size_t count) {
  if (count == 0) count = 1;
  HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
  return GCWT_OK;
}

template <typename T>
int upload_vec(T** p, const std::vector<T>& v, hipStream_t st) {
  int rc = dev_alloc(p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
  return GCWT_OK;
}

void free_dev(gcwt_plan* p) {
  auto fr = [](auto*& q) { if (q) { (void)hipFree((void*)q); q = nullptr; } };
  fr(p->d_y); fr(p->d_tw64); fr(p->d_x); fr(p->d_xr); fr(p->d_xb); fr(p->d_xs); fr(p->d_probe); fr(p->d_amps); fr(p->d_z); fr(p->d_hfull); fr(p->d_bank); fr(p->d_gain); fr(p->d_gain_lv); fr(p->d_half_tw); fr(p->d_psi); fr(p->d_psi_tail); fr(p->d_psi_lit); fr(p->d_tw4096);
  fr(p->d_tw256); fr(p->d_level_tw); fr(p->d_sums); fr(p->d_scale_list); fr(p->d_scale_aux); fr(p->d_interp_coef); fr(p->d_bank_sc); fr(p->d_direct_sc); fr(p->d_bc_h); fr(p->d_bc_rows); fr(p->d_bc_x); fr(p->d_bc_tw);
  fr(p->d_in);
  if (p->d_out) { (void)hipFree(p->d_out); p->d_out = nullptr; }
  if (p->h_pred) { (void)hipHostFree(p->h_pred); p->h_pred = nullptr; }
  fr(p->d_hist); fr(p->d_bands); fr(p->d_pred); fr(p->d_scale_level); fr(p->d_scale_length); fr(p->d_run_mask);
  for (gcwt_plan*& sub : p->sub_plan)
    if (sub) { gcwt_plan_destroy(sub); sub = nullptr; }
  for (auto& kv : p->hfull_cache) (void)hipFree(kv.second);
  p->hfull_cache.clear();
  p->hfull_cache_bytes = 0;
  p->host_out.release();
  for (auto& e : p->ep_dev) { fr(e.items); fr(e.levels); fr(e.items7); fr(e.items7w); fr(e.items7n); fr(e.levels7); fr(e.items_i); fr(e.levels_i); fr(e.items_p[0]); fr(e.items_p[1]); fr(e.levels_p); fr(e.pred_levels); }
  p->ep_dev.clear();
  if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
  for (auto e : p->ev_pool) (void)hipEventDestroy(e);
  p->ev_pool.clear();
  if (p->stream) { (void)hipStreamDestroy(p->stream); p->stream = nullptr; }
  for (auto& q : p->aux) if (q) { (void)hipStreamDestroy(q); q = nullptr; }
  if (p->det_stream) { (void)hipStreamDestroy(p->det_stream); p->det_stream = nullptr; }
  p->uploaded = false;
}

int get_event(gcwt_plan* p, hipEvent_t* e) {
  if (p->ev_used == p->ev_pool.size()) {
    hipEvent_t n;
    HIP_TRY(hipEventCreate(&n));
    p->ev_pool.push_back(n);
  }
  *e = p->ev_pool[p->ev_used++];
  return GCWT_OK;
}

// Which synthesis kernel makes a level: the interpolating one when the planner designed it
// (amplitude / power, R >= 16), else k_synth7 when the block layout allows, else the 16-column
// fallback.  GHOSTCWT_SYNTH16=1 sends everything to the fallback (A/B tests).
enum LevelKernel { LK_SYNTH16 = 0, LK_SYNTH7 = 7, LK_INTERP = 9 };
// an interpolated level goes to the pipelined kernel (synthp.hip) when its layout fits: two phases per scale, a
// lane's sub-sample positions fixed by the lane (I <= 256), whole lane-tasks per block (hop R a multiple of 4 I)
inline bool level_pipelined(const gcwt_plan* p, const LevelPlan& lp) {
  return p->use_synthp && lp.interp_q == 2 && lp.interp_factor >= 4 && lp.interp_factor <= kSynthpMaxFactor &&
         lp.scales.size() <= 256;
}
inline LevelKernel level_kernel(const gcwt_plan* p, const LevelPlan& lp) {
  // (the 16-column kernel knows nothing of a band shift: shifted levels keep k_synth7 whatever the option says)
  if (p->use_synth16 && lp.band_shift == 0) return LK_SYNTH16;
  if (lp.interp_q > 0) return LK_INTERP;
  // (a shifted band is built into k_synth7 and k_synthi only; k_synth7<WIDE> takes its long halos)
  return lp.fast || (lp.band_shift > 0 && lp.scales.size() <= 256) ? LK_SYNTH7 : LK_SYNTH16;
}

// RAII-less span helper: begin/end record events on the stage's stream when profiling
struct SpanGuard {
  gcwt_plan* p;
  int stage;
  hipEvent_t a = nullptr;
  int rc = GCWT_OK;
  bool on = false;
  SpanGuard(gcwt_plan* p_, int st) : p(p_), stage(st) {
    // (level 2: the events around the other stages cost the step they measure 0.19 ms of 13.5 -- tools/step_gap.py)
    on = p->profiling == 1 || (p->profiling == 2 && (st == ST_SYNTH || st == ST_INTERP));
    if (!on) return;
    rc = get_event(p, &a);
    if (rc == GCWT_OK && hipEventRecord(a, p->cur ? p->cur : p->stream) != hipSuccess) rc = GCWT_ERR_HIP;
  }
  int end() {
    if (!on || rc) return rc;
    hipEvent_t b;
    rc = get_event(p, &b);
    if (rc) return rc;
    if (hipEventRecord(b, p->cur ? p->cur : p->stream) != hipSuccess) return GCWT_ERR_HIP;
    p->spans.push_back({stage, a, b});
    return GCWT_OK;
  }
};


}  // namespace

extern "C" {

int gcwt_abi_version(void) { return GCWT_ABI_VERSION; }

const char* gcwt_last_error(void) { return g_err.c_str(); }

int gcwt_device_count(int* count) {
  if (!count) return set_err(GCWT_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return set_err(GCWT_ERR_NO_DEVICE, hipGetErrorString(e)); }
  *count = n;
  return GCWT_OK;
}

int gcwt_device_name(int device, char* buf, size_t buflen) {
  if (!buf || buflen == 0) return set_err(GCWT_ERR_INVALID, "buf is NULL");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  // some driver stacks leave prop.name empty: say so rather than print a blank
  snprintf(buf, buflen, "%s (%s, %d CUs, %.0f GiB)", prop.name[0] ? prop.name : "AMD GPU (unnamed by the driver)",
           prop.gcnArchName, prop.multiProcessorCount, (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
  return GCWT_OK;
}

int gcwt_device_pci_bus_id(int device, char* buf, size_t buflen) {
  if (!buf || buflen < 16) return set_err(GCWT_ERR_INVALID, "buffer of at least 16 bytes needed");
  HIP_TRY(hipDeviceGetPCIBusId(buf, (int)buflen, device));
  return GCWT_OK;
}

int gcwt_set_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    (void)hipGetLastError();
    return set_err(GCWT_ERR_NO_DEVICE, "no HIP device: libghostcwt needs an AMD GPU (gfx950); there is no CPU path");
  }
  if (device < 0 || device >= n) return set_err(GCWT_ERR_NO_DEVICE, "no such device (gcwt_device_count tells how many there are)");
  HIP_TRY(hipSetDevice(device));
  return GCWT_OK;
}
int gcwt_current_device(int* device) {
  if (!device) return set_err(GCWT_ERR_INVALID, "NULL argument");
  HIP_TRY(hipGetDevice(device));
  return GCWT_OK;
}

int gcwt_device_memory(size_t* free_bytes, size_t* total_bytes) {
  if (!free_bytes || !total_bytes) return set_err(GCWT_ERR_INVALID, "NULL argument");
  HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
  return GCWT_OK;
}
int gcwt_device_malloc(void** ptr, size_t bytes) { HIP_TRY(hipMalloc(ptr, bytes ? bytes : 1)); return GCWT_OK; }
int gcwt_device_free(void* ptr) { HIP_TRY(hipFree(ptr)); return GCWT_OK; }
int gcwt_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return GCWT_OK;
}
int gcwt_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return GCWT_OK;
}
int gcwt_device_memset(void* dst, int value, size_t bytes) {
  HIP_TRY(hipMemset(dst, value, bytes)); return GCWT_OK;
}
int gcwt_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return GCWT_OK; }
int gcwt_host_alloc(void** ptr, size_t bytes) {
  if (!ptr) return set_err(GCWT_ERR_INVALID, "NULL argument");
  HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return GCWT_OK;
}
int gcwt_host_free(void* ptr) { HIP_TRY(hipHostFree(ptr)); return GCWT_OK; }
int gcwt_rows_to_host(const float* d_src, int64_t src_pitch, int64_t n_rows, int64_t row_elems, void* dst,
                      int64_t dst_pitch, int flags) {
  return guarded([&] {
    if (!d_src || !dst) return set_err(GCWT_ERR_INVALID, "NULL argument");
    if (n_rows < 0 || row_elems < 0 || src_pitch < row_elems || dst_pitch < row_elems)
      return set_err(GCWT_ERR_INVALID, "bad rectangle (rows, elements per row, pitches)");
    if (!(flags & GCWT_HOST_PINNED) && dst_pitch != row_elems)
      return set_err(GCWT_ERR_INVALID, "a destination that is not page-locked takes dense rows (dst_pitch = row_elems)");
    HIP_TRY(rows_to_host(d_src, src_pitch, n_rows, row_elems, dst, dst_pitch, (flags & GCWT_OUT_F64) != 0,
                         (flags & GCWT_HOST_PINNED) != 0));
    return (int)GCWT_OK;
  });
}

static int gcwt_plan_create_impl(gcwt_plan** out, const gcwt_params* params) {
  if (!out || !params) return set_err(GCWT_ERR_INVALID, "NULL argument");
  *out = nullptr;
  // (held by a unique_ptr until it is handed over: build_host_plan may throw std::bad_alloc, which guarded()
  // turns into GCWT_ERR_NOMEM after the partly built plan is gone)
  std::unique_ptr<gcwt_plan> owner(new (std::nothrow) gcwt_plan());
  gcwt_plan* p = owner.get();
  if (!p) return set_err(GCWT_ERR_NOMEM, "out of host memory");
  std::string err;
  int rc = build_host_plan(*params, &p->hp, &err);
  if (rc != GCWT_OK) return set_err(rc, err);
  // the plan keeps its own copies of the arrays
  p->hp.prm.freqs_hz = p->hp.freqs.data();
  p->hp.prm.epoch_bounds = p->hp.bounds.data();
  p->hp.prm.n_epochs = (int32_t)(p->hp.bounds.size() / 2);
  p->device = params->device;
  p->high_precision = p->hp.high_precision;
  // options (options.h): read once, here; an execute never looks at them or at the environment
  p->use_synth16 = option_or("synth16", 0) == 1;
  p->synth_cols = option_or("synth_cols", 32) == 16 ? 16 : 32;
  p->synth7_narrow_r = (int)option_or("synth7_narrow_r", 2);
  p->fuse_blocks = option_or("fuse_blocks", 1) != 0;
  p->prune_inputs = option_or("prune_inputs", 1) != 0;
  p->fast_fft = option_or("slow_fft", 0) == 0;
  p->level_streams = option_or("level_streams", 1) != 0;
  if (option_is_set("interp_grid")) p->interp_grid = option_or("interp_grid", 0) != 0;
  p->synth_streams = option_or("synth_streams", 0) != 0;
  p->fold_mean = option_or("fold_mean", 1) != 0;
  p->interp_lgnb = (int)option_or("interp_lgnb", -1);
  p->use_synthp = kMeasureBuild && option_or("synthp", 0) != 0;   // (the kernel exists in the measure build only)
  p->synthp_lgnb = (int)option_or("synthp_lgnb", -1);
  p->synthp_help = (int)option_or("synthp_help", -1);
  p->use_graphs = option_or("graphs", 1) != 0;
  p->synth7_order = (int)option_or("synth7_order", 0);
  p->auto_threshold = 1e-9f * (float)option_or("auto_threshold_ppb", 1500);
  p->kappa_eps = 1e-9f * (float)option_or("auto_kappa_ppb", 160);
  p->oob_tol = 1e-12f * (float)option_or("auto_oob_ppt", 25000);
  p->fullband_group = (int)option_or("fullband_group", 0);
  p->fullband4 = option_or("fullband4", 1) != 0;
  p->synth_kernel = kMeasureBuild && option_or("synth_kernel", 7) == 8 ? 8 : 7;
  p->drop_stores = kMeasureBuild && option_is_set("synth_drop_stores") ? (int)std::max<long long>(1, option_or("synth_drop_stores", 1)) : 0;
  p->clock_probe = kMeasureBuild && option_is_set("clock_probe");
  for (const auto& s : p->hp.scales)
    if (s.method == GCWT_SCALE_DIRECT) p->max_direct_len = std::max(p->max_direct_len, s.length);
  *out = owner.release();
  return GCWT_OK;
}

void gcwt_plan_destroy(gcwt_plan* plan) {
  if (!plan) return;
  if (plan->uploaded || plan->stream) {
    if (plan->device >= 0) (void)hipSetDevice(plan->device);
    free_dev(plan);
  }
  delete plan;
}

int gcwt_plan_get_info(const gcwt_plan* plan, gcwt_plan_info* info) {
  if (!plan || !info) return set_err(GCWT_ERR_INVALID, "NULL argument");
  const HostPlan& hp = plan->hp;
  info->abi_version = GCWT_ABI_VERSION;
  info->n_levels = (int32_t)hp.levels.size();
  info->n_direct = hp.n_direct;
  info->n_spectral = (int32_t)hp.scales.size() - hp.n_direct - hp.n_fullband - hp.n_blockconv;
  info->n_fullband = hp.n_fullband;
  info->n_blockconv = hp.n_blockconv;
  info->reserved = 0;
  info->n_interp = 0;
  for (const auto& l : hp.levels)
    if (level_kernel(plan, l) == LK_INTERP) info->n_interp += (int32_t)l.scales.size();
  info->block = hp.block;
  int r = 1;
  for (const auto& l : hp.levels) r = std::max(r, l.decimation);
  info->max_decimation = r;
  info->fft_length = hp.max_p;
  // the planner's figure (X, x_R, XB, Z, bank, tables) plus what this file allocates on top of it
  {
    const int64_t C = hp.prm.n_channels, slots = C * hp.max_batch;
    int64_t extra = 0, xs = 0, y = 0;
    for (const EpochPlan& ep : hp.epochs) {
      for (size_t l = 0; l < hp.levels.size(); ++l)
        if (hp.levels[l].band_shift > 0) xs = std::max<int64_t>(xs, ep.lv[l].m);
      const int rows = ep.p1 >= 4 && hp.n_fullband == 0 ? ep.p1 / 2 + 1 : ep.p1;
      y = std::max<int64_t>(y, (int64_t)rows * kRowLen);
    }
    extra += 8 * 3 * slots * xs;                                        // shifted-band slices, one per level stream
    if (hp.high_precision && plan->fast_fft && hp.n_direct + hp.n_blockconv < hp.prm.n_freqs)
      extra += 16 * (slots * y + 8192);   // float64 intermediate + twiddles (what gcwt_plan_upload allocates: d_y, d_tw64)
    extra += 8 * (2 * (int64_t)hp.direct_total + (int64_t)hp.n_direct);   // literal taps and tap sums beside the running sums
    int64_t listed = 0;
    for (const LevelPlan& lp : hp.levels) listed += (int64_t)lp.scales.size();
    extra += 4 * (listed + 8) * 256 + 8 * 256 * (int64_t)hp.levels.size();   // gain rows in list order, half-sample twiddles
    extra += 8 * (int64_t)channel_sum_doubles((size_t)C) + 4 * (int64_t)hp.interp_coef.size();
    if (hp.high_precision && !hp.exact_only) {             // the detector's band sums and predictions (precision auto / high)
      int64_t max_rows = 1;
      for (const EpochPlan& ep : hp.epochs) max_rows = std::max<int64_t>(max_rows, ep.p1);
      extra += 4 * (slots * (max_rows * kRowBands + kSpecBands) + (int64_t)hp.prm.n_freqs * (int64_t)hp.epochs.size());
    }
    if (hp.n_fullband > 0)
      extra += std::min<int64_t>(plan->uploaded ? plan->fullband_cache_cap : kFullbandCacheBytes,
                                 8 * (int64_t)hp.n_fullband * hp.max_p * (int64_t)hp.epochs.size());
    info->workspace_bytes = hp.workspace_bytes + extra;
    // precision = auto: the exact sub-plans a recording with in-band interference made this plan create (at most two,
    // kept for later executes) are device memory of this plan too
    for (const gcwt_plan* sub : plan->sub_plan) {
      gcwt_plan_info si{};
      if (sub && gcwt_plan_get_info(sub, &si) == GCWT_OK) info->workspace_bytes += si.workspace_bytes;
    }
  }
  info->out_bytes = (int64_t)hp.out_elem_bytes * hp.prm.n_channels * hp.prm.n_freqs * hp.prm.n_samples;
  return GCWT_OK;
}

int gcwt_plan_scale_info(const gcwt_plan* plan, int32_t* method, int32_t* decimation, int32_t* halo,
                         int32_t* hop, int64_t* length) {
  if (!plan) return set_err(GCWT_ERR_INVALID, "NULL plan");
  const HostPlan& hp = plan->hp;
  for (size_t i = 0; i < hp.scales.size(); ++i) {
    const ScalePlan& s = hp.scales[i];
    if (method) method[i] = s.method;
    if (decimation) decimation[i] = s.method == GCWT_SCALE_SPECTRAL ? s.decimation : 1;
    if (halo) halo[i] = s.level >= 0 ? hp.levels[s.level].halo : 0;
    if (hop) hop[i] = s.level >= 0 ? hp.levels[s.level].hop : 0;
    if (length) length[i] = s.length;
  }
  return GCWT_OK;
}

int gcwt_plan_scale_support(const gcwt_plan* plan, double* theta_hi, double* support, int32_t* n_bins) {
  if (!plan) return set_err(GCWT_ERR_INVALID, "NULL plan");
  const HostPlan& hp = plan->hp;
  for (size_t i = 0; i < hp.scales.size(); ++i) {
    const ScalePlan& s = hp.scales[i];
    if (theta_hi) theta_hi[i] = s.theta_hi;
    if (support) support[i] = s.support;
    if (n_bins) n_bins[i] = s.n_bins;
  }
  return GCWT_OK;
}

int gcwt_plan_set_profiling(gcwt_plan* plan, int enabled) {
  if (!plan) return set_err(GCWT_ERR_INVALID, "NULL plan");
  plan->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
  return GCWT_OK;
}

int gcwt_plan_set_row_pitch(gcwt_plan* plan, int64_t pitch_samples) {
  if (!plan || pitch_samples < 0) return set_err(GCWT_ERR_INVALID, "bad row pitch");
  plan->row_pitch = pitch_samples;
  return GCWT_OK;
}

static int gcwt_plan_upload_impl(gcwt_plan* p) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  if (p->uploaded) return GCWT_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return set_err(GCWT_ERR_NO_DEVICE,
                   "no HIP device: libghostcwt needs an AMD GPU (gfx950); there is no CPU path");
  if (p->device >= 0) HIP_TRY(hipSetDevice(p->device));
  const HostPlan& hp = p->hp;
  const int64_t C = hp.prm.n_channels;
  const int S = hp.prm.n_freqs, B = hp.block;
  int rc;
  // option cu_count (measurements: profiles/r06_bound.md): the plan's streams may use that many of the CUs only.  The
  // mask's bits are dealt round-robin over the XCDs by the driver, so its first n bits are n / 8 CUs of each.
  auto make_stream = [&](hipStream_t* q) -> hipError_t {
    const int n_cu = (int)option_or("cu_count", 0);
    if (n_cu <= 0) return hipStreamCreateWithFlags(q, hipStreamNonBlocking);
    uint32_t mask[16] = {};
    for (int i = 0; i < n_cu && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
    return hipExtStreamCreateWithCUMask(q, 16, mask);
  };
  HIP_TRY(make_stream(&p->stream));
  for (auto& q : p->aux) HIP_TRY(make_stream(&q));
  HIP_TRY(make_stream(&p->det_stream));
  auto bail = [&](int code) { free_dev(p); return code; };
  bool he_sync_tables = false;

  const bool any_fft = hp.n_direct + hp.n_blockconv < S;   // spectral or full-band scales: they share X
  if (any_fft) {
    // one workspace slot per (segment of a batch, channel)
    const int64_t slots = C * hp.max_batch;
    if ((rc = dev_alloc(&p->d_x, (size_t)(slots * hp.max_p_store)))) return bail(rc);
    if (p->high_precision && p->fast_fft) {
      // rows 0 .. P1/2 of the intermediate (all of them when full-band scales read the whole spectrum)
      for (const EpochPlan& ep : hp.epochs) {
        const int rows = ep.p1 >= 4 && hp.n_fullband == 0 ? ep.p1 / 2 + 1 : ep.p1;
        p->y_stride = std::max<int64_t>(p->y_stride, (int64_t)rows * kRowLen);
      }
      if ((rc = dev_alloc(&p->d_y, (size_t)(slots * p->y_stride)))) return bail(rc);
      std::vector<double2> tw(8192);
      fwd64_fill_tables(tw.data());
      if ((rc = upload_vec(&p->d_tw64, tw, p->stream))) return bail(rc);
      he_sync_tables = true;
    }
    if ((rc = dev_alloc(&p->d_xr, (size_t)(slots * hp.max_xr)))) return bail(rc);
    if ((rc = dev_alloc(&p->d_xb, (size_t)(slots * hp.max_xb)))) return bail(rc);
    for (const EpochPlan& ep : hp.epochs)
      for (size_t l = 0; l < hp.levels.size(); ++l)
        if (hp.levels[l].band_shift > 0) p->xs_stride = std::max(p->xs_stride, ep.lv[l].m);
    // one slice buffer per stream the level passes run on (run_pipeline: three streams)
    if (p->xs_stride > 0 && (rc = dev_alloc(&p->d_xs, (size_t)(3 * slots * p->xs_stride)))) return bail(rc);
    if (hp.n_fullband > 0) {
      p->z_sets = std::min(hp.n_fullband, kFullbandSet);   // that many scales share a pass over X
      p->z_half = slots * hp.max_p;
      if ((rc = dev_alloc(&p->d_z, (size_t)(p->z_sets * p->z_half)))) return bail(rc);
      if ((rc = dev_alloc(&p->d_hfull, (size_t)(p->z_sets * hp.max_p)))) return bail(rc);
    }
  }
  if (he_sync_tables && hipStreamSynchronize(p->stream) != hipSuccess)      // the host table went out of scope
    return bail(set_err(GCWT_ERR_HIP, "twiddle upload"));
  if ((rc = upload_vec(&p->d_amps, hp.amps, p->stream))) return bail(rc);
  if (p->clock_probe) {
    if ((rc = dev_alloc(&p->d_probe, 8))) return bail(rc);
    hipError_t he0 = hipMemsetAsync(p->d_probe, 0, 64, p->stream);
    if (he0 != hipSuccess) return bail(hip_err(he0, "probe reset"));
  }
  if ((rc = dev_alloc(&p->d_bank, (size_t)S * B))) return bail(rc);
  if ((rc = dev_alloc(&p->d_gain, (size_t)S * B))) return bail(rc);
  if ((rc = dev_alloc(&p->d_psi, (size_t)hp.direct_total))) return bail(rc);
  if ((rc = dev_alloc(&p->d_psi_lit, (size_t)hp.direct_total))) return bail(rc);
  if ((rc = dev_alloc(&p->d_psi_tail, (size_t)hp.n_direct))) return bail(rc);
  {   // zero taps in front of every kernel and behind it up to a multiple of 8 (+8): k_direct reads whole groups
    hipError_t hz = hipMemsetAsync(p->d_psi, 0, sizeof(float2) * (size_t)std::max<int64_t>(1, hp.direct_total), p->stream);
    if (hz != hipSuccess) return bail(hip_err(hz, "psi reset"));
  }
  if ((rc = dev_alloc(&p->d_sums, channel_sum_doubles((size_t)C)))) return bail(rc);
  {   // results and partial sums of k_channel_sum
    hipError_t hz = hipMemsetAsync(p->d_sums, 0, sizeof(double) * channel_sum_doubles((size_t)C), p->stream);
    if (hz != hipSuccess) return bail(hip_err(hz, "sums reset"));
  }

  // tables, computed in double on the host
  std::vector<float2> tw4096(kRowLen / 2), tw256(256), ltw((size_t)hp.level_twiddle_total);
  for (int j = 0; j < kRowLen / 2; ++j) {
    double a = -2.0 * M_PI * j / kRowLen;
    tw4096[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  for (int q = 0; q < 256; ++q) {
    double a = 2.0 * M_PI * q / 256.0;
    tw256[q] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  for (const LevelPlan& lp : hp.levels) {
    const int64_t n = (int64_t)kSynthCols * lp.decimation;
    for (int64_t q = 0; q < n; ++q) {
      double a = 2.0 * M_PI * (double)q / ((double)B * lp.decimation);
      ltw[lp.twiddle_offset + q] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
  }
  if ((rc = upload_vec(&p->d_tw4096, tw4096, p->stream))) return bail(rc);
  if ((rc = upload_vec(&p->d_tw256, tw256, p->stream))) return bail(rc);
  if ((rc = upload_vec(&p->d_level_tw, ltw, p->stream))) return bail(rc);

  std::vector<BankScale> bsc(S);
  std::vector<DirectScale> dsc(hp.n_direct);
  for (int i = 0; i < S; ++i) {
    const ScalePlan& s = hp.scales[i];
    bsc[i] = {s.omega, s.half_delay, s.length, s.amp_offset, s.bin_lo, s.n_bins, s.decimation,
              s.method == GCWT_SCALE_SPECTRAL ? 1 : 0,
              s.method == GCWT_SCALE_SPECTRAL ? hp.levels[s.level].band_shift : 0, 0};
    if (s.method == GCWT_SCALE_DIRECT)
      dsc[s.direct_index] = {s.omega, s.length, s.amp_offset, s.direct_offset, i, s.bin_lo, s.n_bins,
                              direct_front_pad(s.length)};
  }
  if ((rc = upload_vec(&p->d_bank_sc, bsc, p->stream))) return bail(rc);
  if ((rc = upload_vec(&p->d_direct_sc, dsc, p->stream))) return bail(rc);
  if (hp.n_blockconv > 0 || hp.n_fullband > 0) {
    // W_4096^(+(t + 16 j) a) at [256 j + 16 t + a]: the middle twiddles of the 4096-point inverse transforms of
    // k_bc_scales and k_fullband_rows as each thread meets them (the values tw4096_at<+1> gives)
    std::vector<float2> twt(kRowLen);
    for (int j = 0; j < 16; ++j)
      for (int tid = 0; tid < 256; ++tid) {
        const int idx = (((tid >> 4) + 16 * j) * (tid & 15)) & (kRowLen - 1);
        float2 w = tw4096[idx & 2047];
        if (idx & 2048) w = make_float2(-w.x, -w.y);
        twt[256 * j + tid] = make_float2(w.x, -w.y);
      }
    if ((rc = upload_vec(&p->d_bc_tw, twt, p->stream))) return bail(rc);
    HIP_TRY(hipStreamSynchronize(p->stream));         // `twt` goes out of scope
  }
  if (hp.n_blockconv > 0) {
    std::vector<int32_t> rows(hp.bc_order.begin(), hp.bc_order.end());
    if ((rc = upload_vec(&p->d_bc_rows, rows, p->stream))) return bail(rc);
    if ((rc = dev_alloc(&p->d_bc_h, (size_t)hp.n_blockconv * kRowLen))) return bail(rc);
    if ((rc = dev_alloc(&p->d_bc_x, (size_t)(hp.bc_chunk_blocks * C) * kRowLen))) return bail(rc);
    if (!p->d_tw64) {
      std::vector<double2> tw(8192);
      fwd64_fill_tables(tw.data());
      if ((rc = upload_vec(&p->d_tw64, tw, p->stream))) return bail(rc);
      HIP_TRY(hipStreamSynchronize(p->stream));       // `tw` goes out of scope
    }
  }

  std::vector<int32_t> scale_list, scale_aux;
  std::vector<int> scale_off(hp.levels.size());
  std::vector<int> n_plain(hp.levels.size());
  std::vector<float2> half_tw(hp.levels.size() * 256);
  for (size_t l = 0; l < hp.levels.size(); ++l) {
    scale_off[l] = (int)scale_list.size();
    // odd kernel lengths (no half-sample delay, real filter) first, even ones after
    for (int sidx : hp.levels[l].scales)
      if (hp.scales[sidx].half_delay == 0.0) scale_list.push_back(sidx);
    n_plain[l] = (int)scale_list.size() - scale_off[l];
    for (int sidx : hp.levels[l].scales)
      if (hp.scales[sidx].half_delay != 0.0) scale_list.push_back(sidx);
    for (int k = 0; k < 256; ++k) {
      const double a = -M_PI * (k - hp.levels[l].band_shift) / (256.0 * hp.levels[l].decimation);
      half_tw[l * 256 + k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
  }
  for (int sidx : scale_list)
    scale_aux.push_back(hp.scales[sidx].demod_bin | (hp.scales[sidx].half_delay != 0.0 ? 1 << 16 : 0));
  if ((rc = upload_vec(&p->d_half_tw, half_tw, p->stream))) return bail(rc);
  if ((rc = upload_vec(&p->d_scale_list, scale_list, p->stream))) return bail(rc);
  if ((rc = upload_vec(&p->d_scale_aux, scale_aux, p->stream))) return bail(rc);
  if ((rc = upload_vec(&p->d_interp_coef, hp.interp_coef, p->stream))) return bail(rc);
  p->n_listed = (int)scale_list.size();
  {   // k_synth7 reads whole chunks of 8 rows: 8 rows of zeros behind the last level's
    const size_t n = ((size_t)p->n_listed + 8) * 256;
    if ((rc = dev_alloc(&p->d_gain_lv, n))) return bail(rc);
    hipError_t hz = hipMemsetAsync(p->d_gain_lv, 0, sizeof(float) * n, p->stream);
    if (hz != hipSuccess) return bail(hip_err(hz, "gain rows reset"));
  }
  p->ep_dev.resize(hp.epochs.size());
  for (size_t e = 0; e < hp.epochs.size(); ++e) {
    const EpochPlan& ep = hp.epochs[e];
    if (ep.batch_count == 0) continue;     // shares the tables of its batch's first segment
    // each level goes to the production kernel when its layout allows (16 <= halo <= 32, at
    // most 256 scales), else to the 16-column kernel; GHOSTCWT_SYNTH16=1 sends everything there
    std::vector<SynthItemDev> items;
    for (size_t i = 0; i < ep.items.size(); ++i)
      if (level_kernel(p, hp.levels[ep.items[i].level]) == LK_SYNTH16)
        items.push_back({ep.items[i].level, ep.items[i].scale, ep.items[i].blk0, ep.items[i].nblk});
    p->ep_dev[e].n_items = (int)items.size();
    std::vector<SynthLevelDev> lv(hp.levels.size());
    for (size_t l = 0; l < lv.size(); ++l)
      lv[l] = {hp.levels[l].decimation, hp.levels[l].hop, hp.levels[l].halo, ep.lv[l].blk_lo,
               ep.lv[l].xb_offset, hp.levels[l].twiddle_offset};
    if ((rc = upload_vec(&p->ep_dev[e].items, items, p->stream))) return bail(rc);
    if ((rc = upload_vec(&p->ep_dev[e].levels, lv, p->stream))) return bail(rc);
    std::vector<Synth7Item> items7, items7w, items7n;
    std::vector<Synth7Level> lv7(hp.levels.size());
    for (size_t l = 0; l < lv7.size(); ++l) {
      const LevelPlan& lp = hp.levels[l];
      int lg = 0;
      while ((1 << lg) < lp.decimation) ++lg;
      lv7[l] = {lp.decimation, lg, lp.hop, lp.halo, ep.lv[l].nblk, (int32_t)lp.scales.size(),
                scale_off[l], ep.lv[l].blk_lo, n_plain[l], (int32_t)(l * 256), lp.band_shift, 0, ep.lv[l].xb_offset,
                lp.twiddle_offset, ep.lv[l].xr_offset, ep.lv[l].m - 1};
      const bool narrow = lp.halo <= 48 && lp.decimation <= p->synth7_narrow_r && p->synth_cols == 32;
      const int cols = narrow ? 16 : p->synth_cols;
      const int bpb = std::max(1, cols / lp.decimation);
      const int n_rtiles = std::max(1, lp.decimation / cols);
      if (level_kernel(p, lp) != LK_SYNTH7) continue;
      for (int b0 = 0; b0 < ep.lv[l].nblk; b0 += bpb)
        for (int rt = 0; rt < n_rtiles; ++rt) (lp.halo > 48 ? items7w : narrow ? items7n : items7).push_back({(int32_t)l, b0, rt, 0});
    }
    // order of the k_synth7 items (option synth7_order, A/B runs): 0 as listed (levels in plan order, R = 2 first),
    // 1 reversed, 2 the levels' items dealt in turn
    if (p->synth7_order == 1) std::reverse(items7.begin(), items7.end());
    else if (p->synth7_order == 2) {
      std::vector<std::vector<Synth7Item>> by(hp.levels.size());
      for (const auto& it : items7) by[(size_t)it.level].push_back(it);
      std::vector<Synth7Item> mixed;
      std::vector<double> pos(by.size(), 0.0);
      size_t left = items7.size();
      while (left) {                                   // always the level that is furthest behind its share
        size_t best = by.size(); double frac = 2.0;
        for (size_t l = 0; l < by.size(); ++l)
          if (pos[l] < (double)by[l].size() && pos[l] / (double)by[l].size() < frac) { frac = pos[l] / (double)by[l].size(); best = l; }
        mixed.push_back(by[best][(size_t)pos[best]]);
        pos[best] += 1.0; --left;
      }
      items7.swap(mixed);
    }
    p->ep_dev[e].n_items7 = (int)items7.size();
    p->ep_dev[e].n_items7w = (int)items7w.size();
    p->ep_dev[e].n_items7n = (int)items7n.size();
    if ((rc = upload_vec(&p->ep_dev[e].items7n, items7n, p->stream))) return bail(rc);
    if ((rc = upload_vec(&p->ep_dev[e].items7, items7, p->stream))) return bail(rc);
    if ((rc = upload_vec(&p->ep_dev[e].items7w, items7w, p->stream))) return bail(rc);
    if ((rc = upload_vec(&p->ep_dev[e].levels7, lv7, p->stream))) return bail(rc);
    // interpolated levels: one workgroup per block, the longest-running (largest R) first
    std::vector<SynthiItem> items_i;
    std::vector<std::pair<int, int>> pending;          // (level, log2 blocks per workgroup)
    std::vector<SynthiLevel> lvi(hp.levels.size());
    std::vector<int> order;
    std::vector<int> order_p;                            // ... the ones the pipelined kernel takes
    for (size_t l = 0; l < hp.levels.size(); ++l)
      if (level_kernel(p, hp.levels[l]) == LK_INTERP)
        (level_pipelined(p, hp.levels[l]) ? order_p : order).push_back((int)l);
    std::sort(order.begin(), order.end(),
              [&](int x, int y) { return hp.levels[x].decimation > hp.levels[y].decimation; });
    std::sort(order_p.begin(), order_p.end(),
              [&](int x, int y) { return hp.levels[x].decimation > hp.levels[y].decimation; });
    for (int l : order) {
      const LevelPlan& lp = hp.levels[l];
      int lgq = 0;
      while ((1 << lgq) < lp.interp_q) ++lgq;
      // Blocks per workgroup: a workgroup's prologue (its blocks' spectra, 16 threads each) takes
      // about as long whatever their number, and a block of a low decimation is little work:
      // two blocks per workgroup up to R = 32, one from R = 64 up (measured per level,
      // profiles/r03_synth_study.md; the 16 columns of a pass are blocks x scales x phases) -- with two phases per scale;
      // with four (R >= 32 since round 5) two blocks leave two scales per pass and one block is the better cut (R = 32:
      // 1.56 against 1.63 - 1.66 ms)
      int lgnb = lp.decimation <= 32 && lp.interp_q <= 2 ? 1 : 0;
      if (p->interp_lgnb >= 0) lgnb = std::min(p->interp_lgnb, 2);
      while (lgnb > 0 && (lp.interp_q << lgnb) > kInterpMaxPhases) --lgnb;
      // what k_synthi's indexing assumes (synthi.hip); the planner guarantees it
      if (lp.interp_q < 2 || lp.interp_q > kInterpMaxPhases || (lp.interp_q & (lp.interp_q - 1)) ||
          lp.interp_factor * lp.interp_q != lp.decimation || lp.interp_factor < 4 ||
          lp.interp_factor > kInterpMaxFactor || lp.scales.size() > 256 || lp.halo < 16 ||
          lp.hop != hp.block - 2 * lp.halo || lp.hop < 1 || hp.block != 256 || (lp.interp_taps != 6 && lp.interp_taps != 8))
        return bail(set_err(GCWT_ERR_INVALID, "internal: interpolated level outside the kernel's limits"));
      lvi[l] = {lp.decimation, lp.interp_q, lgq, lp.interp_factor, lp.hop, lp.halo, ep.lv[l].nblk,
                (int32_t)lp.scales.size(), scale_off[l], ep.lv[l].blk_lo, lgnb, lp.interp_taps, lp.twiddle_offset,
                ep.lv[l].xr_offset, ep.lv[l].m - 1, lp.coef_offset};
      pending.push_back({l, lgnb});
    }
    // A workgroup walks all the scales of its blocks unless that is too much for one: the passes
    // of a level's walk (kInterpCols / (q nb) scales each) are cut into runs of at most
    // `target` output bytes, and where a single pass is more than that (high decimations: one
    // pass of two scales of a block is 14 MB at R = 8192) its wave-tasks are shared out over
    // several workgroups, each of which repeats the pass's transforms (17 q / R of the work).
    // `target` is halved until the launch has a few rounds of workgroups per CU.  The list is
    // ordered largest first, so that what is still running when the launch ends is the small items.  A run's prologue -- its blocks' spectra -- costs ~8 us.
    {
      const int64_t n_ch_slots = (int64_t)hp.prm.n_channels * std::max(1, ep.batch_count);
      int64_t target = (int64_t)2 << 20;
      std::vector<int64_t> item_bytes;
      for (int attempt = 0;; ++attempt) {
        items_i.clear();
        item_bytes.clear();
        for (const auto& pl : pending) {
          const LevelPlan& lp = hp.levels[pl.first];
          const int nb = 1 << pl.second, ns = kInterpCols / (lp.interp_q * nb);
          const int n_pass = ((int)lp.scales.size() + ns - 1) / ns;
          const int64_t pass_bytes = (int64_t)ns * nb * lp.hop * lp.decimation * 4;
          const int run = (int)std::max<int64_t>(1, std::min<int64_t>(n_pass, target / std::max<int64_t>(1, pass_bytes)));
          const int wps = (lp.hop * (lp.decimation >> 2) + 63) >> 6;     // wave-tasks per (block, scale): synthi.hip
          const int quads = (wps + 3) / 4;
          const int parts = run > 1 ? 1 : (int)std::min<int64_t>(quads, (pass_bytes + target - 1) / target);
          for (int b0 = 0; b0 < ep.lv[pl.first].nblk; b0 += nb)
            for (int p0 = 0; p0 < n_pass; p0 += run)
              for (int part = 0; part < parts; ++part) {
                const int lo = 4 * (int)((int64_t)quads * part / parts);
                const int hi = std::min(wps, 4 * (int)((int64_t)quads * (part + 1) / parts));
                const int np = std::min(run, n_pass - p0);
                items_i.push_back({(int32_t)pl.first, b0, p0, np, lo, hi});
                item_bytes.push_back((int64_t)np * pass_bytes * (hi - lo) / std::max(1, wps));
              }
        }
        if ((int64_t)items_i.size() * n_ch_slots >= 3072 || target <= ((int64_t)1 << 19) || attempt > 6) break;
        target >>= 1;
      }
      std::vector<size_t> order(items_i.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return item_bytes[x] > item_bytes[y]; });
      std::vector<SynthiItem> sorted(items_i.size());
      for (size_t i = 0; i < order.size(); ++i) sorted[i] = items_i[order[i]];
      items_i.swap(sorted);
    }
    p->ep_dev[e].n_items_i = (int)items_i.size();
    if ((rc = upload_vec(&p->ep_dev[e].items_i, items_i, p->stream))) return bail(rc);
    if ((rc = upload_vec(&p->ep_dev[e].levels_i, lvi, p->stream))) return bail(rc);
    // pipelined kernel: a workgroup owns nb consecutive blocks and walks the level's scales 4 / nb at a time (a
    // round: 8 columns, 4 z slots).  Blocks per workgroup by decimation, so that a (round, scale) run is at least a
    // few dozen wave-tasks of 256 samples for the six consumer waves: four blocks up to R = 16, two at R = 32, one
    // beyond.  A run of rounds is cut at `target` bytes of rows like k_synthi's passes; largest items first.
    {
      std::vector<SynthpLevel> lvp(hp.levels.size());
      std::vector<SynthpItem> items_p[2];
      std::vector<int64_t> bytes_p[2];
      const int64_t n_ch_slots = (int64_t)hp.prm.n_channels * std::max(1, ep.batch_count);
      for (int l : order_p) {
        const LevelPlan& lp = hp.levels[l];
        int lgnb = lp.decimation <= 16 ? 2 : lp.decimation <= 32 ? 1 : 0;
        if (p->synthp_lgnb >= 0) lgnb = std::min(p->synthp_lgnb, 2);
        if (lp.interp_factor * 2 != lp.decimation || lp.halo < 16 || lp.hop != hp.block - 2 * lp.halo || lp.hop < 1 ||
            hp.block != 256 || (lp.hop * lp.decimation) % (4 * lp.interp_factor) != 0)
          return bail(set_err(GCWT_ERR_INVALID, "internal: pipelined level outside the kernel's limits"));
        // what the two producer waves take of a round's tasks after making the next round's z (about 4.5 tasks' worth
        // of instructions each): all eight waves done together when (T - h) / 6 = 4.5 + h / 2
        const double tasks = 4.0 * lp.hop * lp.decimation / 256.0;
        int help = (int)std::lround(128.0 * std::max(0.0, (tasks - 27.0) / (4.0 * tasks)));
        if (p->synthp_help >= 0) help = std::min(p->synthp_help, 64);
        lvp[l] = {lp.decimation, lp.interp_factor, lp.hop, lp.halo, ep.lv[l].nblk, (int32_t)lp.scales.size(),
                  scale_off[l], ep.lv[l].blk_lo, lgnb, n_plain[l], (int32_t)(l * 256), help, lp.twiddle_offset,
                  ep.lv[l].xr_offset, ep.lv[l].m - 1, lp.coef_offset};
      }
      int64_t target = (int64_t)2 << 20;
      for (int attempt = 0;; ++attempt) {
        for (int k = 0; k < 2; ++k) { items_p[k].clear(); bytes_p[k].clear(); }
        for (int l : order_p) {
          const LevelPlan& lp = hp.levels[l];
          const int nb = 1 << lvp[l].log2nb, ns = kSynthpSlots / nb;
          const int n_rounds = ((int)lp.scales.size() + ns - 1) / ns;
          const int64_t round_bytes = (int64_t)ns * nb * lp.hop * lp.decimation * 4;
          const int run = (int)std::max<int64_t>(1, std::min<int64_t>(n_rounds, target / std::max<int64_t>(1, round_bytes)));
          const int k = lp.interp_factor == 4 ? 1 : 0;
          for (int b0 = 0; b0 < ep.lv[l].nblk; b0 += nb)
            for (int r0 = 0; r0 < n_rounds; r0 += run) {
              const int nr = std::min(run, n_rounds - r0);
              items_p[k].push_back({(int32_t)l, b0, r0, nr});
              bytes_p[k].push_back((int64_t)nr * round_bytes);
            }
        }
        if ((int64_t)(items_p[0].size() + items_p[1].size()) * n_ch_slots >= 2048 || target <= ((int64_t)1 << 19) || attempt > 6) break;
        target >>= 1;
      }
      for (int k = 0; k < 2; ++k) {
        std::vector<size_t> ord(items_p[k].size());
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return bytes_p[k][x] > bytes_p[k][y]; });
        std::vector<SynthpItem> sorted(items_p[k].size());
        for (size_t i = 0; i < ord.size(); ++i) sorted[i] = items_p[k][ord[i]];
        p->ep_dev[e].n_items_p[k] = (int)sorted.size();
        if ((rc = upload_vec(&p->ep_dev[e].items_p[k], sorted, p->stream))) return bail(rc);
      }
      if ((rc = upload_vec(&p->ep_dev[e].levels_p, lvp, p->stream))) return bail(rc);
    }
  }

  // precision = auto / high: the detector's tables (detect.hip)
  {
    bool long_mode = false, any_level = false;
    for (const EpochPlan& ep : hp.epochs) long_mode = long_mode || ep.long_a > 1;
    std::vector<int32_t> scale_level((size_t)S, -1);
    for (int i = 0; i < S; ++i)
      if (hp.scales[i].method == GCWT_SCALE_SPECTRAL && hp.scales[i].level >= 0) { scale_level[(size_t)i] = hp.scales[i].level; any_level = true; }
    p->detect = hp.high_precision && !hp.exact_only && p->d_y && !long_mode && any_level;
    if (p->detect) {
      const int64_t slots = (int64_t)C * hp.max_batch;
      int64_t max_rows = 1;                              // rows of the forward row pass: run_pipeline's rows_a
      for (const EpochPlan& ep : hp.epochs)
        max_rows = std::max<int64_t>(max_rows, ep.p1);
      if ((rc = dev_alloc(&p->d_hist, (size_t)(slots * max_rows * kRowBands)))) return bail(rc);
      if ((rc = dev_alloc(&p->d_bands, (size_t)(slots * kSpecBands)))) return bail(rc);
      if ((rc = dev_alloc(&p->d_pred, (size_t)S * (size_t)C * hp.epochs.size()))) return bail(rc);
      if ((rc = upload_vec(&p->d_scale_level, scale_level, p->stream))) return bail(rc);
      std::vector<int32_t> scale_length((size_t)S);
      for (int i = 0; i < S; ++i) scale_length[(size_t)i] = (int32_t)std::min<int64_t>(hp.scales[i].length, (int64_t)1 << 30);
      if ((rc = upload_vec(&p->d_scale_length, scale_length, p->stream))) return bail(rc);
      for (size_t e = 0; e < hp.epochs.size(); ++e) {
        const EpochPlan& ep = hp.epochs[e];
        if (ep.batch_count == 0) continue;
        std::vector<PredLevel> pl(hp.levels.size());
        for (size_t l = 0; l < hp.levels.size(); ++l) {
          const LevelPlan& lp = hp.levels[l];
          const LevelPlan& own = hp.levels[lp.xr_owner >= 0 ? (size_t)lp.xr_owner : l];   // whose x_R the level reads
          pl[l] = {lp.scales.empty() ? 0 : lp.decimation, lp.band_shift, 0.f, 0.f};
          if (own.taper_hi > 0.0 && own.band_shift == 0) {                                 // run_pipeline: RowTaper
            const double k1 = own.taper_hi * (double)ep.p / (2.0 * M_PI), k0 = 0.5 * k1;
            if (k1 - k0 >= 1.0) { pl[l].k0 = (float)k0; pl[l].k1 = (float)k1; }
          }
        }
        if ((rc = upload_vec(&p->ep_dev[e].pred_levels, pl, p->stream))) return bail(rc);
      }
      p->last_pred.assign((size_t)S, 0.f);
      HIP_TRY(hipHostMalloc((void**)&p->h_pred, sizeof(float) * (size_t)S * (size_t)C * hp.epochs.size(), hipHostMallocDefault));
      HIP_TRY(hipStreamSynchronize(p->stream));       // the vectors of this block go out of scope
    }
  }

  hipError_t he = launch_build_bank(p->d_bank, p->d_gain, p->d_bank_sc, p->d_amps, S, B, p->stream);
  if (he != hipSuccess) return bail(hip_err(he, "build_bank"));
  he = launch_scale_windows(p->d_gain, p->d_scale_list, p->n_listed, (float)hp.band_tol, p->d_gain_lv,
                            p->prune_inputs, p->stream);
  if (he != hipSuccess) return bail(hip_err(he, "scale_windows"));
  he = launch_build_direct(p->d_psi, p->d_direct_sc, hp.n_direct, p->max_direct_len, p->d_amps, p->d_psi_tail,
                           p->d_psi_lit, p->stream);
  if (he != hipSuccess) return bail(hip_err(he, "build_direct"));
  for (int k = 0; k < hp.n_blockconv; ++k) {      // H on the 4096-point grid of a block, over 4096 (exact.hip)
    he = launch_fullband_filter(p->d_bc_h + (int64_t)k * kRowLen, p->d_bank_sc, hp.bc_order[k], p->d_amps, 1, p->stream);
    if (he != hipSuccess) return bail(hip_err(he, "blockconv responses"));
  }
  he = hipStreamSynchronize(p->stream);  // host vectors above go out of scope
  if (he != hipSuccess) return bail(hip_err(he, "plan upload"));
  if (hp.n_fullband > 0) {
    // full-band responses kept across executes: a quarter of what is free now (after the workspace), at most
    // 16 GiB; option fullband_cache_mb sets it (0: none are kept).  Nothing is allocated for it here: a response
    // enters the cache when its scale is first computed, and a failed allocation closes the cache.
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    p->fullband_cache_cap = std::min<int64_t>(kFullbandCacheBytes, (int64_t)(free_b / 4));
    if (p->is_sub_plan) p->fullband_cache_cap /= 2;      // two of them may live inside one parent
    if (option_is_set("fullband_cache_mb")) p->fullband_cache_cap = std::max<long long>(0, option_or("fullband_cache_mb", 0)) << 20;
  }
  p->uploaded = true;
  return GCWT_OK;
}

// Computes samples [r0, r1) of every (channel, scale) row into rows of row_len samples
// (column = sample - r0).  The full transform is r0 = 0, r1 = N, row_len = N.
static int run_pipeline(gcwt_plan* p, const float* dx, float* dout, int64_t r0, int64_t r1,
                        int64_t row_len, bool reuse_means) {
  const HostPlan& hp = p->hp;
  const int C = hp.prm.n_channels, S = hp.prm.n_freqs;
  const int64_t N = hp.prm.n_samples;
  const int mode = hp.prm.out_mode;
  const int elem = mode == GCWT_OUT_COMPLEX_C64 ? 2 : 1;
  const double inv_n = 1.0 / (double)N;
  hipStream_t st = p->stream;
  hipError_t he;
  p->cur = nullptr;            // (an earlier call may have failed between a fork and its join)
#define RUN(stage_id, call)                         \
  do {                                              \
    SpanGuard sg_(p, stage_id);                     \
    if (sg_.rc) return sg_.rc;                      \
    he = (call);                                    \
    if (he != hipSuccess) return hip_err(he, #call);\
    int rc_ = sg_.end();                            \
    if (rc_) return rc_;                            \
  } while (0)

  // The channel means (transforms.py:142-143).  When the plan is one segment that is the whole recording (no epochs cut
  // out, no time blocks with faded edges, no interleaved transforms) and every scale reads the spectrum, the forward column
  // pass sums the samples it reads anyway and the row pass takes the mean's transform out of its input (fwd64.hip):
  // the recording is read once.  A block request of such a plan runs the same forward side, so it gives the same bits.
  bool fold = false;
  if (p->fold_mean && p->d_y && hp.epochs.size() == 1 && hp.n_direct == 0 && hp.n_blockconv == 0) {
    const EpochPlan& m = hp.epochs[0];
    fold = m.start == 0 && m.ne == N && m.lead == 0 && m.ramp_lo == 0 && m.ramp_hi == 0 && m.long_a == 1 &&
           std::max(1, m.batch_count) == 1;
  }
  p->last_fold = fold;
  if (!reuse_means && !fold) RUN(ST_MEAN, launch_channel_sum(dx, N, C, p->d_sums, st));
  if (p->detect) {
    he = hipMemsetAsync(p->d_pred, 0, sizeof(float) * (size_t)S * (size_t)C * hp.epochs.size(), st);
    if (he != hipSuccess) return hip_err(he, "predictions reset");
  }

  // samples outside every epoch are zero (transforms.py:185): one launch per gap, or one
  // fill of the whole result when there are many of them (a masked run writes into rows that are finished but for
  // the marked scales: their gaps are zero already)
  const unsigned char* const hmask = p->run_mask;
  const unsigned char* const dmask = hmask ? p->d_run_mask : nullptr;
  if (!hmask) {
    std::vector<std::pair<int64_t, int64_t>> eps, gaps;
    for (size_t i = 0; i + 1 < hp.bounds.size(); i += 2) eps.push_back({hp.bounds[i], hp.bounds[i + 1]});
    std::sort(eps.begin(), eps.end());
    int64_t cursor = r0;
    for (size_t i = 0; i <= eps.size(); ++i) {
      const int64_t gap_end = std::min(r1, i < eps.size() ? eps[i].first : N);
      if (gap_end > cursor) gaps.push_back({cursor, gap_end});
      if (i < eps.size()) cursor = std::max(cursor, std::min(r1, eps[i].second));
    }
    if (gaps.size() > 8) {
      he = hipMemsetAsync(dout, 0, sizeof(float) * (size_t)elem * (size_t)row_len * (size_t)C * (size_t)S, st);
      if (he != hipSuccess) return hip_err(he, "zero fill");
    } else {
      for (const auto& g : gaps) {
        he = launch_zero_range(dout, row_len * elem, (int64_t)C * S, (g.first - r0) * elem,
                               (g.second - g.first) * elem, st);
        if (he != hipSuccess) return hip_err(he, "zero_range");
      }
    }
  }

  bool any_fft = hp.n_direct + hp.n_blockconv < S;
  if (hmask) {                       // a masked run needs the spectrum only for a marked scale that reads it
    any_fft = false;
    for (int i = 0; i < S; ++i)
      any_fft = any_fft || (hmask[i] && hp.scales[i].method != GCWT_SCALE_DIRECT && hp.scales[i].method != GCWT_SCALE_BLOCKCONV);
  }
  const bool fast_fft = p->fast_fft;
  for (size_t e0 = 0; any_fft && e0 < hp.epochs.size();) {
    // one batch: segments e0 .. e0 + count - 1 share the FFT length and the level grids;
    // those with something to write in [r0, r1) become extra sets of "channels"
    const EpochPlan& ep = hp.epochs[e0];
    const size_t count = (size_t)std::max(1, ep.batch_count);
    SegIn sin{};
    SegOut sout{};
    PredSegs psegs{};
    int nb = 0;
    for (size_t i = 0; i < count; ++i) {
      const EpochPlan& m = hp.epochs[e0 + i];
      // output window of the segment, segment-local: its core, cut to the range
      const int64_t w_lo = std::max(m.core0, r0) - m.start, w_hi = std::min(m.core1, r1) - m.start;
      if (w_hi <= w_lo) continue;
      psegs.seg[nb] = (int32_t)(e0 + i);
      sin.x_off[nb] = m.start;
      sin.n_valid[nb] = m.ne;
      sin.n_lead[nb] = m.lead;
      sin.ramp_lo[nb] = m.ramp_lo;
      sin.ramp_hi[nb] = m.ramp_hi;
      sout.seg_col[nb] = m.start - r0;
      sout.w_lo[nb] = w_lo;
      sout.w_hi[nb] = w_hi;
      ++nb;
    }
    e0 += count;
    if (nb == 0) continue;
    sin.n_channels = sout.n_channels = psegs.n_channels = C;
    const int slots = C * nb;
    // P, P1: the STORED spectrum (rows of 4096 bins); Pt: the segment's true FFT length.  They differ in long mode
    // only (planner.h, EpochPlan::long_a: Pt = A P, the spectrum's low half combined from A interleaved transforms)
    const int64_t P = ep.p_store, Pt = ep.p;
    const int P1 = ep.p1, A = ep.long_a;
    // The input is real, so rows k1 and P1 - k1 of the k1-major spectrum mirror each other:
    // the fast path builds rows 0 .. P1/2 only and writes the rest as their reflections.
    // Full-band scales read every bin, so a plan that has any builds the whole spectrum.
    const bool hermitian = fast_fft && P1 >= 4 && hp.n_fullband == 0;
    const int rows_a = hermitian ? P1 / 2 + 1 : P1;
    // forward FFT, pass A: FFT over n1 (stride 4096) of x[4096 n1 + n2], twiddle W_P^{-n2 k1}
    if (p->d_y) {
      // precision = high: both passes in float64, the spectrum rounded to float32 per bin (fwd64.hip)
      for (int a = 0; a < A; ++a) {
        RUN(ST_FWD, launch_fwd64_cols(dx, p->d_y, P1, N, p->y_stride, P, p->d_tw64, p->d_sums, inv_n, sin, nb,
                                      rows_a, st, A, a, fold));
        if (fold) RUN(ST_MEAN, launch_channel_sum_final(p->d_sums, C, fold_parts(P1), st));
        RUN(ST_FWD, launch_fwd64_rows(p->d_y, p->d_x, rows_a, p->y_stride, P, p->d_tw64, slots,
                                      hp.n_fullband > 0 ? kRowLen : kRowLen / 2, hermitian ? P1 : 0, st, a, A, Pt,
                                      p->detect ? p->d_hist : nullptr, P1, fold ? p->d_sums : nullptr, inv_n, ep.ne, P1));
      }
      // precision = auto / high: the row pass left the band energies of every row of the spectrum (fwd64.hip:
      // row_band_sums); on a stream of its own, beside the level passes and the synthesis, they are added up and turned
      // into what the float32 stages will cost each scale (detect.hip); joined at the end of the batch
      if (p->detect) {
        hipEvent_t bands_done;
        int rc_ = get_event(p, &bands_done);
        if (rc_) return rc_;
        he = hipEventRecord(bands_done, st);
        if (he == hipSuccess) he = hipStreamWaitEvent(p->det_stream, bands_done, 0);
        if (he != hipSuccess) return hip_err(he, "detector fork");
        he = launch_band_sums(p->d_hist, P1, p->d_bands, slots, p->det_stream);
        if (he != hipSuccess) return hip_err(he, "launch_band_sums");
        he = launch_precision_predict(p->d_bands, p->d_gain, p->d_scale_level, p->d_scale_length, p->ep_dev[ep.batch_first].pred_levels, S,
                                      (int)hp.levels.size(), (double)Pt, p->kappa_eps, p->oob_tol, p->d_pred, nullptr, nullptr, slots, psegs,
                                      p->det_stream);
        if (he != hipSuccess) return hip_err(he, "launch_precision_predict");
        p->last_batch_slots = slots; p->last_batch = ep.batch_first; p->last_pt = (double)Pt; p->last_rows = P1;
      }
    } else if (A > 1) {
      return set_err(GCWT_ERR_UNSUPPORTED, "internal: long mode without the float64 forward transform");
    } else {
    RUN(ST_FWD, launch_fft_cols_batch(dx, p->d_x, P1, kRowLen, N, P, P1 > 1 ? P : 0, p->d_tw4096,
                                      fast_fft ? p->d_tw256 : nullptr, p->d_sums, inv_n, sin, nb, st,
                                      rows_a));
    // pass B: rows over n2 -> X~[k1][k2] = X[k1 + P1 k2]; only X[k < P/2] is ever read
    RUN(ST_FWD, launch_fft_rows(-1, p->d_x, p->d_x, kRowLen, rows_a, kRowLen, kRowLen, P, P, 0,
                                p->d_tw4096, fast_fft ? p->d_tw256 : nullptr, 1.0f, slots, st,
                                hp.n_fullband > 0 ? kRowLen : kRowLen / 2, hermitian ? P1 : 0));
    }
    // the production synthesis kernel computes its blocks' spectra itself (no XB pass)
    const bool fused_blocks = p->fuse_blocks && p->synth_kernel == 7 && !p->use_synth16;
    // The levels are independent once the spectrum is there (rows pass -> column pass per
    // level, disjoint x_R); run one after the other they leave 10 us between launches and the
    // small ones (R >= 16: grids that do not fill the chip) cost 0.2 ms.  Three streams, the
    // plan's own being one of them, share them by estimated time -- two passes over the level's
    // P/R samples per slot at the rate such passes reach, plus what two launches cost however
    // small they are (config 5's levels are all of the second kind and end up three per stream;
    // the headline's R = 2 level is half of all the work and keeps a stream to itself) -- and
    // join before the synthesis.
    const bool side = p->level_streams && hp.levels.size() > 2;
    std::vector<int> stream_of(hp.levels.size(), 0);
    if (side) {
      double load[3] = {0.0, 0.0, 0.0};
      std::vector<size_t> order;
      for (size_t l = 0; l < hp.levels.size(); ++l)
        if (hp.levels[l].xr_owner == (int)l) order.push_back(l);
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ep.lv[a].m > ep.lv[b].m; });
      for (size_t l : order) {
        const int k = (int)(std::min_element(load, load + 3) - load);
        stream_of[l] = k;
        load[k] += 40e-6 + 32.0 * (double)ep.lv[l].m * (double)slots / 4.5e12;   // seconds
      }
      for (size_t l = 0; l < hp.levels.size(); ++l) stream_of[l] = stream_of[hp.levels[l].xr_owner];
    }
    bool forked[2] = {false, false};
    hipEvent_t spectrum_ready = nullptr;
    if (side) {
      int rc_ = get_event(p, &spectrum_ready);
      if (rc_) return rc_;
      he = hipEventRecord(spectrum_ready, st);
      if (he != hipSuccess) return hip_err(he, "hipEventRecord");
    }
    for (size_t l = 0; l < hp.levels.size(); ++l) {
      const LevelPlan& lp = hp.levels[l];
      const EpochLevel& el = ep.lv[l];
      float2* xr = p->d_xr + el.xr_offset;
      hipStream_t ls = st;
      RowTaper taper;                      // precision = high: the slice loses what lies below every scale's band
      if (p->high_precision && lp.taper_hi > 0.0 && lp.band_shift == 0) {
        const double k1 = lp.taper_hi * (double)Pt / (2.0 * M_PI), k0 = 0.5 * k1;
        if (k1 - k0 >= 1.0) { taper.p1 = P1; taper.k0 = (float)k0; taper.inv_width = (float)(1.0 / (k1 - k0)); }
      }
      if (side) {                          // level (and whoever shares its x_R) -> its own stream
        const int k = stream_of[l];
        ls = k == 0 ? st : p->aux[k - 1];
        if (k > 0 && !forked[k - 1]) {
          he = hipStreamWaitEvent(ls, spectrum_ready, 0);
          if (he != hipSuccess) return hip_err(he, "hipStreamWaitEvent");
          forked[k - 1] = true;
        }
        p->cur = ls;
      }
      if (lp.xr_owner != (int)l) {
        // x_R of this decimation was made for the level that owns it (an earlier one)
      } else if (lp.decimation / A <= kMaxTwoPassDecimation) {
        const int Q = kRowLen * A / lp.decimation;        // M = P1 Q samples: decimation R / A of the stored spectrum
        const float2* src = p->d_x;
        int64_t src_row = kRowLen, src_cstride = P;
        if (lp.band_shift > 0) {
          // the level's band starts band_shift bins below zero: its M spectrum samples are gathered
          // (negative frequencies as conjugates) into a slice buffer first, one per stream
          const int64_t U = (int64_t)lp.band_shift * el.m / hp.block;     // a multiple of P1 (planner: steps of R / 16 bins)
          if (U % P1 != 0 || U / P1 > Q) return set_err(GCWT_ERR_INVALID, "internal: band shift does not slice the spectrum");
          float2* xs = p->d_xs + (int64_t)stream_of[l] * slots * p->xs_stride;
          RUN(ST_DECIM, launch_shift_gather(p->d_x, xs, P1, Q, (int)(U / P1), kRowLen, P, p->xs_stride, slots, ls));
          src = xs; src_row = Q; src_cstride = p->xs_stride;
        }
        // x_R[Q m1 + m2] = sum_{j1} e^{2 pi i j1 m1/P1} e^{2 pi i j1 m2/M} sum_{j2} X~[j1][j2] e^{2 pi i j2 m2/Q}
        RUN(ST_DECIM, launch_fft_rows(+1, src, xr, Q, P1, src_row, Q, src_cstride, hp.max_xr,
                                      P1 > 1 ? el.m : 0, p->d_tw4096,
                                      fast_fft ? p->d_tw256 : nullptr, 1.0f, slots, ls, 0, 0, taper));
        if (P1 > 1)
          RUN(ST_DECIM, launch_fft_cols(+1, false, xr, xr, P1, Q, hp.max_xr, hp.max_xr, 0,
                                        p->d_tw4096, fast_fft ? p->d_tw256 : nullptr, p->d_sums,
                                        inv_n, 0, slots, ls));
      } else {
        // M = P/R <= 8192: rows j1 < n1 of X~, q leading entries each
        const int n1 = (int)std::min<int64_t>(P1, el.m);
        const int q = (int)(el.m / n1);
        RUN(ST_DECIM, launch_level_small(p->d_x, xr, n1, q, kRowLen, P, hp.max_xr, p->d_tw4096, slots, ls, taper));
      }
      const float scale = (float)(1.0 / ((double)hp.block * (double)Pt));
      const LevelKernel lk = level_kernel(p, lp);
      if (lk == LK_SYNTH16 || (lk == LK_SYNTH7 && !fused_blocks))     // the kernels that read XB
        RUN(ST_BLOCK, launch_block_fft(xr, p->d_xb + el.xb_offset, el.m, lp.hop, lp.halo, el.blk_lo,
                                       el.nblk, hp.max_xr, hp.max_xb, p->d_tw256, scale, slots, ls));
    }
    if (side) {
      p->cur = nullptr;
      for (int k = 0; k < 2; ++k) {
        if (!forked[k]) continue;
        hipEvent_t done;
        int rc_ = get_event(p, &done);
        if (rc_) return rc_;
        he = hipEventRecord(done, p->aux[k]);
        if (he == hipSuccess) he = hipStreamWaitEvent(st, done, 0);
        if (he != hipSuccess) return hip_err(he, "level streams join");
      }
    }
    const EpochDev& dev = p->ep_dev[ep.batch_first];
    // The interpolating kernel is launched first; with GHOSTCWT_SYNTH_STREAMS=1 on a stream of its own, k_synth7 beside it:
    // the two share the CUs (store-bound workgroups next to arithmetic-bound ones) and each
    // fills the other's tail.  Both only read what the level passes left and write disjoint rows.
    hipStream_t si = st;
    if (dev.n_items_i > 0) {
      const bool beside = p->synth_streams && (dev.n_items7 > 0 || dev.n_items7w > 0 || dev.n_items7n > 0 || dev.n_items > 0);
      if (beside) {
        hipEvent_t levels_done;
        int rc_ = get_event(p, &levels_done);
        if (rc_) return rc_;
        he = hipEventRecord(levels_done, st);
        if (he == hipSuccess) he = hipStreamWaitEvent(p->aux[0], levels_done, 0);
        if (he != hipSuccess) return hip_err(he, "synthesis fork");
        si = p->aux[0];
      }
      SynthiArgs ai{};
      ai.tw256 = p->d_tw256;
      ai.level_tw = p->d_level_tw;
      ai.items = dev.items_i;
      ai.levels = dev.levels_i;
      ai.scale_list = p->d_scale_list;
      ai.scale_aux = p->d_scale_aux;
      ai.gain = p->d_gain;
      ai.coef = p->d_interp_coef;
      ai.out = dout;
      ai.row_len = row_len;
      ai.xr = p->d_xr;
      ai.xr_cstride = hp.max_xr;
      ai.xb_scale = (float)(1.0 / ((double)hp.block * (double)Pt));
      ai.n_scales = S;
      // Which index runs fastest in the grid decides what is in flight together: the same few items
      // of every channel, or many items of one channel.  Measured (profiles/r03_synth_study.md 12): with
      // 24 channel slots (config 5) channels-fastest is 3 % faster; with 128 it is as fast in most
      // processes and 8 % slower in some (the placement of the 51 GB of rows decides), items-fastest
      // never is.
      ai.channels_fastest = p->interp_grid >= 0 ? p->interp_grid : (slots <= 32 ? 1 : 0);
      if (dev.n_items_i > 65535) ai.channels_fastest = 0;       // (grid.y is 16 bits wide)
      ai.seg = sout;
      p->cur = si;
      RUN(ST_INTERP, launch_synthi(mode, ai, dev.n_items_i, slots, si));
      p->cur = nullptr;
    }
    for (int k = 0; k < 2; ++k) {
      if (dev.n_items_p[k] == 0) continue;
      SynthpArgs ap{};
      ap.tw256 = p->d_tw256;
      ap.level_tw = p->d_level_tw;
      ap.items = dev.items_p[k];
      ap.levels = dev.levels_p;
      ap.scale_list = p->d_scale_list;
      ap.scale_aux = p->d_scale_aux;
      ap.gain_lv = p->d_gain_lv;
      ap.level_half_tw = p->d_half_tw;
      ap.coef = p->d_interp_coef;
      ap.out = dout;
      ap.row_len = row_len;
      ap.xr = p->d_xr;
      ap.xr_cstride = hp.max_xr;
      ap.xb_scale = (float)(1.0 / ((double)hp.block * (double)Pt));
      ap.n_scales = S;
      ap.channels_fastest = p->interp_grid >= 0 ? p->interp_grid : (slots <= 32 ? 1 : 0);
      if (dev.n_items_p[k] > 65535) ap.channels_fastest = 0;     // (grid.y is 16 bits wide)
      ap.flags = 0;
      ap.seg = sout;
      p->cur = si;
#ifdef GCWT_MEASURE
      RUN(ST_INTERP, launch_synthp(mode, ap, dev.n_items_p[k], slots, k == 1, si));
#else
      return set_err(GCWT_ERR_INVALID, "internal: a level planned for the measure build's pipelined kernel");
#endif
      p->cur = nullptr;
    }
    if (dev.n_items > 0) {
      SynthArgs a{};
      a.xb = p->d_xb;
      a.bank = p->d_bank;
      a.tw256 = p->d_tw256;
      a.level_tw = p->d_level_tw;
      a.items = dev.items;
      a.levels = dev.levels;
      a.out = dout;
      a.xb_cstride = hp.max_xb;
      a.row_len = row_len;
      a.n_scales = S;
      a.seg = sout;
      RUN(ST_SYNTH, launch_synth(mode, a, dev.n_items, slots, st));
    }
    for (int variant = 0; variant < 3; ++variant) {          // 16 columns (R = 2) first, then the 32-column lists
      const int wide = variant == 2;
      const int cols7 = variant == 0 ? 16 : p->synth_cols;
      const int n7 = variant == 0 ? dev.n_items7n : wide ? dev.n_items7w : dev.n_items7;
      if (n7 == 0) continue;
      Synth7Args a7{};
      a7.xb = p->d_xb;
      a7.bank = p->d_bank;
      a7.tw256 = p->d_tw256;
      a7.level_tw = p->d_level_tw;
      a7.items = variant == 0 ? dev.items7n : wide ? dev.items7w : dev.items7;
      a7.levels = dev.levels7;
      a7.scale_list = p->d_scale_list;
      a7.gain = p->d_gain;
      a7.gain_lv = p->d_gain_lv;
      a7.level_half_tw = p->d_half_tw;
      a7.out = dout;
      a7.xb_cstride = hp.max_xb;
      a7.row_len = row_len;
      a7.n_scales = S;
      a7.drop_stores = p->drop_stores;
      a7.seg = sout;
      a7.clock_probe = p->d_probe;
      if (fused_blocks) {
        a7.xr = p->d_xr;
        a7.xr_cstride = hp.max_xr;
        a7.xb_scale = (float)(1.0 / ((double)hp.block * (double)Pt));
      }
#ifdef GCWT_MEASURE
      if (p->synth_kernel == 8 && variant == 1)
        RUN(ST_SYNTH, launch_synth8(mode, p->synth_cols, a7, n7, slots, st));
      else
#endif
        RUN(ST_SYNTH, launch_synth7(mode, cols7, wide != 0, a7, n7, slots, st));
    }
    if (si != st) {                        // join: the batch is done when both kernels are
      hipEvent_t interp_done;
      int rc_ = get_event(p, &interp_done);
      if (rc_) return rc_;
      he = hipEventRecord(interp_done, si);
      if (he == hipSuccess) he = hipStreamWaitEvent(st, interp_done, 0);
      if (he != hipSuccess) return hip_err(he, "synthesis join");
    }
    if (p->detect && p->d_y) {             // join: the batch is done when its predictions are (d_hist is free again)
      hipEvent_t pred_done;
      int rc_ = get_event(p, &pred_done);
      if (rc_) return rc_;
      he = hipEventRecord(pred_done, p->det_stream);
      if (he == hipSuccess) he = hipStreamWaitEvent(st, pred_done, 0);
      if (he != hipSuccess) return hip_err(he, "detector join");
    }
    if (p->profiling && !hp.levels.empty()) p->last.synth_launches++;
    // full-band scales: W = IFFT_P(X H_s) for every slot of the batch, up to four scales per pass over X
    auto response = [&](int i, int scratch, const float2** h_out) -> int {
      const auto cached = p->hfull_cache.find({i, P1});
      if (cached != p->hfull_cache.end()) { *h_out = cached->second; return GCWT_OK; }
      float2* dst = p->d_hfull + (int64_t)scratch * hp.max_p;
      float2* keep = nullptr;
      const int64_t bytes = (int64_t)sizeof(float2) * P;
      if (!p->hfull_cache_full && p->hfull_cache_bytes + bytes <= p->fullband_cache_cap) {
        if (hipMalloc((void**)&keep, (size_t)bytes) == hipSuccess) dst = keep;
        else {                               // no room: compute into the scratch row as before, and stop asking
          (void)hipGetLastError();
          p->hfull_cache_full = true;
        }
      }
      {   // the response enters the cache only once its launch has been accepted: a failed launch must not
          // leave an unfilled buffer behind for later executes to reuse
        SpanGuard sg_(p, ST_FULLBAND);
        int rc_ = sg_.rc;
        if (!rc_) {
          he = launch_fullband_filter(dst, p->d_bank_sc, i, p->d_amps, P1, st);
          if (he != hipSuccess) rc_ = hip_err(he, "launch_fullband_filter");
        }
        if (!rc_) rc_ = sg_.end();
        if (rc_) { if (keep) (void)hipFree(keep); return rc_; }
      }
      if (keep) {
        p->hfull_cache[{i, P1}] = keep;
        p->hfull_cache_bytes += bytes;
      }
      *h_out = dst;
      return GCWT_OK;
    };
    for (int i = 0; i < S && hp.n_fullband > 0;) {
      int member[kFullbandSet], np = 0;
      for (; i < S && np < p->z_sets; ++i)
        if (hp.scales[i].method == GCWT_SCALE_FULLBAND && (!hmask || hmask[i])) member[np++] = i;
      if (np == 0) break;
      FullbandSet set{};
      set.n = np;
      for (int k = 0; k < np; ++k) {
        set.z[k] = p->d_z + (int64_t)k * p->z_half;
        const int rc_ = response(member[k], k, &set.h[k]);
        if (rc_) return rc_;
      }
      if (P1 == 4 && p->fullband4) {       // time blocks of 16 384 samples: both passes in one kernel, no z
        const float2* hh[kFullbandSet];
        int32_t rows_[kFullbandSet];
        for (int k = 0; k < np; ++k) { hh[k] = set.h[k]; rows_[k] = member[k]; }
        RUN(ST_FULLBAND, launch_fullband4(mode, p->d_x, hh, rows_, np, dout, P, p->d_bc_tw, p->d_tw256, S, row_len, sout,
                                          slots, st));
        continue;
      }
      // inverse: rows over k2 of the products X H with the W_P^(k1 n2) twiddle, then columns over k1 ->
      // natural order; the usual FFT lengths store from the column pass's registers (2.9 GB of traffic per
      // scale at the headline shape; product, two passes and a store kernel moved 8)
      RUN(ST_FULLBAND, launch_fullband_rows(p->d_x, set, P1, P, P, p->d_bc_tw, p->d_tw256, slots, st,
                                            p->fullband_group));
      for (int k = 0; k < np; ++k) {
        if (fullband_cols_fused(P1)) {
          RUN(ST_FULLBAND, launch_fullband_cols(mode, set.z[k], dout, P1, P, p->d_tw4096, p->d_tw256, member[k], S,
                                                row_len, sout, slots, st));
        } else {
          if (P1 > 1)
            RUN(ST_FULLBAND, launch_fft_cols(+1, false, set.z[k], set.z[k], P1, kRowLen, P, P, 0, p->d_tw4096,
                                             p->d_tw256, p->d_sums, inv_n, 0, slots, st));
          RUN(ST_FULLBAND, launch_fullband_store(mode, set.z[k], dout, P, member[k], S, row_len, sout, nb, st));
        }
      }
    }
  }
  bool direct_wanted = hp.n_direct > 0;
  if (hmask) {
    direct_wanted = false;
    for (int i = 0; i < S; ++i) direct_wanted = direct_wanted || (hmask[i] && hp.scales[i].method == GCWT_SCALE_DIRECT);
  }
  if (direct_wanted) {
    DirectEpochs eps{};
    eps.n_channels = C;
    int ne = 0;
    auto flush = [&]() -> int {
      if (ne > 0)
        RUN(ST_DIRECT, launch_direct(mode, dx, dout, p->d_psi, p->d_direct_sc, hp.n_direct, p->d_sums,
                                     inv_n, N, S, eps, ne, r0, row_len, p->max_direct_len, p->d_psi_tail, st, dmask));
      ne = 0;
      return GCWT_OK;
    };
    for (size_t i = 0; i + 1 < hp.bounds.size(); i += 2) {
      const int64_t e0 = hp.bounds[i], e1 = hp.bounds[i + 1];
      const int64_t g_lo = std::max(e0, r0), g_hi = std::min(e1, r1);
      if (g_hi <= g_lo) continue;
      eps.epoch_start[ne] = e0;
      eps.epoch_len[ne] = e1 - e0;
      eps.g_lo[ne] = g_lo;
      eps.g_hi[ne] = g_hi;
      // grid.z = epochs * channels of one launch stays within 65535
      if (++ne == kSegBatch || (int64_t)(ne + 1) * C > 65535) { int rc_ = flush(); if (rc_) return rc_; }
    }
    int rc_ = flush();
    if (rc_) return rc_;
  }
  if (hp.n_blockconv > 0) {
    // block convolution: per group of scales, per batch of epochs, per chunk of blocks -- the blocks' spectra,
    // then every scale of the group from them
    for (const HostPlan::BcGroup& g : hp.bc_groups) {
      if (hmask) {                     // masked run: a group none of whose scales is marked makes no block spectra either
        bool wanted = false;
        for (int k = g.first; k < g.first + g.count; ++k) wanted = wanted || hmask[hp.bc_order[(size_t)k]];
        if (!wanted) continue;
      }
      BcBlocks bl{};
      bl.n_channels = C;
      bl.hop = g.hop;
      bl.back = g.back;
      bl.ramp = g.ramp;
      int ne = 0;
      auto flush = [&]() -> int {
        bl.n_epochs = ne;
        const int total = ne > 0 ? bl.blk_first[ne] : 0;
        const int chunk = (int)std::max<int64_t>(2, hp.bc_chunk_blocks & ~(int64_t)1);
        for (int b0 = 0; b0 < total; b0 += chunk) {
          const int nblk = std::min(chunk, total - b0);
          RUN(ST_BLOCKCONV, launch_bc_forward(dx, p->d_bc_x, bl, b0, nblk, N, p->d_tw64, p->d_sums, inv_n, st));
          RUN(ST_BLOCKCONV, launch_bc_scales(mode, p->d_bc_x, dout, p->d_bc_h + (int64_t)g.first * kRowLen,
                                             p->d_bc_rows + g.first, g.count, p->d_bc_tw, p->d_tw256, bl, b0, nblk,
                                             S, r0, row_len, st, dmask));
        }
        ne = 0;
        return GCWT_OK;
      };
      for (size_t i = 0; i + 1 < hp.bounds.size(); i += 2) {
        const int64_t e0 = hp.bounds[i], e1 = hp.bounds[i + 1];
        const int64_t g_lo = std::max(e0, r0), g_hi = std::min(e1, r1);
        if (g_hi <= g_lo) continue;
        // whole pairs of blocks, even-aligned in recording time: the forward transform carries two blocks at a
        // time, and which two must not depend on the range asked for (execute_block gives execute's bits)
        const int64_t blocks = (((g_hi - 1) / g.hop) | 1) - ((g_lo / g.hop) & ~(int64_t)1) + 1;
        if (ne > 0 && (int64_t)bl.blk_first[ne] + blocks > (1 << 30)) { int rc_ = flush(); if (rc_) return rc_; }
        if (ne == 0) bl.blk_first[0] = 0;
        bl.epoch_start[ne] = e0;
        bl.epoch_stop[ne] = e1;
        bl.g_lo[ne] = g_lo;
        bl.g_hi[ne] = g_hi;
        bl.blk_first[ne + 1] = bl.blk_first[ne] + (int32_t)blocks;
        if (++ne == kSegBatch) { int rc_ = flush(); if (rc_) return rc_; }
      }
      int rc_ = flush();
      if (rc_) return rc_;
    }
  }
#undef RUN
  return GCWT_OK;
}

static int execute_range(gcwt_plan* p, const void* x, void* out, int64_t r0, int64_t r1, int flags);
static int gcwt_plan_create_impl(gcwt_plan** out, const gcwt_params* params);

// precision = auto: samples [r0, r1) of the scales `over` again (dout's rows start at sample out_r0), by a plan of precision = exact that holds just
// them (made on first use, kept per set of scales), from the same device-resident recording; its dense rows are
// copied over the fast path's.  The sub-plan computes its own channel means (same numbers: same kernel, same x).
static int reroute_scales(gcwt_plan* p, const float* dx, float* dout, int64_t out_r0, int64_t r0, int64_t r1, int64_t row_len,
                          const std::vector<int32_t>& over, int channel) {
  // channel >= 0: that channel alone, by the one-channel sub-plan (a recording with one bad electrode pays for one).
  // The sub-plans hold every scale and a run is masked to `over`: whatever the verdicts of a recording's segments
  // are -- every epoch its own set -- two plans serve them all, and a scale's numbers do not depend on the set it
  // is asked for in (a plan per set cost 35 ms each time a set came back after four others).
  const HostPlan& hp = p->hp;
  const int S = hp.prm.n_freqs;
  const int nch = channel >= 0 ? 1 : hp.prm.n_channels;
  gcwt_plan*& sub = p->sub_plan[channel >= 0 ? 1 : 0];
  if (!sub) {
    std::vector<double> f(hp.freqs.begin(), hp.freqs.end());
    gcwt_params prm = hp.prm;
    prm.freqs_hz = f.data();
    prm.n_freqs = (int32_t)f.size();
    prm.n_channels = nch;
    prm.precision = GCWT_PRECISION_EXACT;
    prm.device = p->device;
    int rc = gcwt_plan_create_impl(&sub, &prm);
    if (rc) { sub = nullptr; return rc; }
    const HostPlan& sh = sub->hp;
    if ((int)sh.scales.size() != S || sh.n_direct + sh.n_blockconv + sh.n_fullband != S) {
      gcwt_plan_destroy(sub); sub = nullptr;
      return set_err(GCWT_ERR_INVALID, "internal: the exact sub-plan keeps a decimated scale");
    }
    sub->is_sub_plan = true;
    if ((rc = gcwt_plan_upload(sub))) { gcwt_plan_destroy(sub); sub = nullptr; return rc; }
    if (hipMalloc((void**)&sub->d_run_mask, (size_t)S) != hipSuccess) {
      (void)hipGetLastError();
      gcwt_plan_destroy(sub); sub = nullptr;
      return set_err(GCWT_ERR_NOMEM, "no device memory for the rerouted scales' mask");
    }
  }
  std::vector<unsigned char> mask((size_t)S, 0);
  for (int32_t i : over) mask[(size_t)i] = 1;
  HIP_TRY(hipMemcpy(sub->d_run_mask, mask.data(), (size_t)S, hipMemcpyHostToDevice));
  const int elem = hp.out_elem_bytes / (int)sizeof(float);
  const int64_t ch0 = channel >= 0 ? channel : 0;
  float* dst = dout + (ch0 * S * row_len + (r0 - out_r0)) * elem;
  sub->row_pitch = row_len;
  sub->run_mask = mask.data();
  const int rc = execute_range(sub, dx + ch0 * hp.prm.n_samples, dst, r0, r1, GCWT_X_ON_DEVICE | GCWT_OUT_ON_DEVICE);
  sub->run_mask = nullptr;
  return rc;
}

static int execute_range(gcwt_plan* p, const void* x, void* out, int64_t r0, int64_t r1, int flags) {
  if ((flags & GCWT_OUT_F64) && (flags & GCWT_OUT_ON_DEVICE))
    return set_err(GCWT_ERR_INVALID, "GCWT_OUT_F64 applies to host output only");
  int rc = gcwt_plan_upload(p);
  if (rc) return rc;
  if (p->device >= 0) HIP_TRY(hipSetDevice(p->device));
  const HostPlan& hp = p->hp;
  const int64_t n_out = r1 - r0;
  const size_t rows = (size_t)hp.prm.n_channels * (size_t)hp.prm.n_freqs;
  // host output: dense for the caller, padded to 32 samples on the device so that every
  // row starts on a 128-byte boundary; device output: the caller's pitch (0 = dense)
  int64_t row_len;
  if (flags & GCWT_OUT_ON_DEVICE) {
    row_len = p->row_pitch ? p->row_pitch : n_out;
    if (row_len < n_out) return set_err(GCWT_ERR_INVALID, "row pitch shorter than the output rows");
  } else {
    row_len = (n_out + 31) & ~(int64_t)31;
  }
  const size_t in_bytes = sizeof(float) * (size_t)hp.prm.n_channels * (size_t)hp.prm.n_samples;
  const size_t out_bytes = hp.out_elem_bytes * rows * (size_t)row_len;
  const float* dx;
  float* dout;
  if (flags & GCWT_X_ON_DEVICE) {
    dx = (const float*)x;
  } else {
    if (p->d_in_bytes < in_bytes) {
      if (p->d_in) { (void)hipFree(p->d_in); p->d_in = nullptr; p->d_in_bytes = 0; }
      HIP_TRY(hipMalloc((void**)&p->d_in, in_bytes));
      p->d_in_bytes = in_bytes;
    }
    HIP_TRY(hipMemcpyAsync(p->d_in, x, in_bytes, hipMemcpyHostToDevice, p->stream));
    dx = p->d_in;
  }
  if (flags & GCWT_OUT_ON_DEVICE) {
    dout = (float*)out;
  } else {
    if (p->d_out_bytes < out_bytes) {
      if (p->d_out) { (void)hipFree(p->d_out); p->d_out = nullptr; p->d_out_bytes = 0; }
      HIP_TRY(hipMalloc(&p->d_out, out_bytes));
      p->d_out_bytes = out_bytes;
    }
    dout = (float*)p->d_out;
  }
  p->ev_used = 0;
  p->spans.clear();
  if (p->profiling) { p->last = gcwt_timings{}; }
  const bool reuse = (flags & GCWT_REUSE_MEANS) && p->have_means;
  // graph replay: device in and out, nothing to time, no full-band scale (its response cache allocates on the way),
  // and a result small enough for the launches to matter (16 M coefficients)
  const bool graphable = !p->graph_failed && !p->profiling && (flags & GCWT_X_ON_DEVICE) && (flags & GCWT_OUT_ON_DEVICE) &&
                         hp.n_fullband == 0 && p->use_graphs && !p->run_mask &&
                         (int64_t)rows * n_out <= ((int64_t)1 << 24);
  const gcwt_plan::GraphKey key{dx, dout, r0, r1, row_len, reuse};
  bool done = false;
  if (graphable && p->graph_exec && key == p->graph_key) {
    HIP_TRY(hipGraphLaunch(p->graph_exec, p->stream));
    done = true;
  } else if (graphable && p->graph_seen_valid && key == p->graph_seen) {
    if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
    hipGraph_t graph = nullptr;
    bool ok = hipStreamBeginCapture(p->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
    if (ok) {
      const int rc_cap = run_pipeline(p, dx, dout, r0, r1, row_len, reuse);
      const hipError_t he_end = hipStreamEndCapture(p->stream, &graph);
      ok = rc_cap == GCWT_OK && he_end == hipSuccess && graph != nullptr;
      if (ok) ok = hipGraphInstantiate(&p->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
    }
    if (ok) {
      p->graph_key = key;
      HIP_TRY(hipGraphLaunch(p->graph_exec, p->stream));
      done = true;
    } else {                                 // whatever did not capture: this plan runs eagerly from here on
      (void)hipGetLastError();
      p->graph_exec = nullptr;
      p->graph_failed = true;
      p->ev_used = 0;
    }
  }
  if (!done) {
    p->graph_seen = key;
    p->graph_seen_valid = true;
    rc = run_pipeline(p, dx, dout, r0, r1, row_len, reuse);
    if (rc) {   // a failure between a fork and its join: nothing may still be running on any of the plan's streams
      (void)hipStreamSynchronize(p->stream);
      for (auto& q : p->aux) (void)hipStreamSynchronize(q);
      if (p->det_stream) (void)hipStreamSynchronize(p->det_stream);
      return rc;
    }
  }
  p->have_means = true;
  if (p->detect) {
    // precision = auto / high: read the predictions; auto makes the scales over the threshold again by the exact paths
    // (a sub-plan with precision = exact for just those scales; its rows replace the fast path's)
    const size_t n_seg = hp.epochs.size(), Sz = (size_t)hp.prm.n_freqs, Cz = (size_t)hp.prm.n_channels;
    HIP_TRY(hipMemcpyAsync(p->h_pred, p->d_pred, sizeof(float) * Sz * Cz * n_seg, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    // The verdict is per segment (epoch or time block) and channel, so that a block request and the whole transform
    // decide alike for the samples they share.  A segment's scales over the threshold (on any of its flagged
    // channels) are made again for the segment's own core: for every channel when half of them or more are flagged,
    // else for the flagged channels one by one (a one-channel sub-plan: a recording with one bad electrode pays for
    // one); runs of neighbouring segments with the same verdict go in one request.
    std::fill(p->last_pred.begin(), p->last_pred.end(), 0.f);
    struct Verdict {
      std::vector<int32_t> scales, channels;
      bool operator==(const Verdict& o) const { return scales == o.scales && channels == o.channels; }
    };
    std::vector<Verdict> vd(n_seg);
    std::vector<char> any(Sz, 0);
    for (size_t e = 0; e < n_seg; ++e) {
      std::vector<char> sc(Sz, 0);
      for (size_t c = 0; c < Cz; ++c) {
        bool flagged = false;
        for (size_t i = 0; i < Sz; ++i) {
          const float v = p->h_pred[(e * Cz + c) * Sz + i];
          p->last_pred[i] = std::max(p->last_pred[i], v);
          if (v > p->auto_threshold) { sc[i] = 1; any[i] = 1; flagged = true; }
        }
        if (flagged) vd[e].channels.push_back((int32_t)c);
      }
      for (size_t i = 0; i < Sz; ++i)
        if (sc[i]) vd[e].scales.push_back((int32_t)i);
      if (2 * vd[e].channels.size() >= Cz) vd[e].channels.clear();            // empty list with scales: every channel
    }
    p->last_worst = 0.f;
    for (float v : p->last_pred) p->last_worst = std::max(p->last_worst, v);
    p->last_rerouted = 0;
    if (hp.auto_precision) {
      // A scale that cannot be made again -- no device memory for the exact sub-plan or its workspace on a small or busy
      // GPU, a layout the exact paths do not take -- keeps the fast path's row: what precision = high returns, complete
      // and finite.  The execute succeeds, the report says so (a negative count) and gcwt_last_error why.
      bool gave_up = false;
      auto soft = [&](int code) {
        if (code != GCWT_ERR_NOMEM && code != GCWT_ERR_UNSUPPORTED) return false;
        (void)hipGetLastError();
        gave_up = true;
        return true;
      };
      for (size_t e = 0; e < n_seg && !gave_up;) {
        size_t e1 = e + 1;
        if (vd[e].scales.empty()) { e = e1; continue; }
        int64_t a = std::max(hp.epochs[e].core0, r0), b = std::min(hp.epochs[e].core1, r1);
        while (e1 < n_seg && vd[e1] == vd[e] && hp.epochs[e1].core0 >= hp.epochs[e1 - 1].core1) {   // (segments come in time order)
          b = std::min(hp.epochs[e1].core1, r1);
          ++e1;
        }
        if (b > a) {
          if (vd[e].channels.empty()) {
            rc = reroute_scales(p, dx, dout, r0, a, b, row_len, vd[e].scales, -1);
            if (rc && !soft(rc)) return rc;
          } else {
            for (int32_t c : vd[e].channels) {
              rc = reroute_scales(p, dx, dout, r0, a, b, row_len, vd[e].scales, c);
              if (rc && !soft(rc)) return rc;
              if (gave_up) break;
            }
          }
        }
        e = e1;
      }
      for (char c : any) p->last_rerouted += c;
      if (gave_up) p->last_rerouted = -p->last_rerouted;
    }
  }
  if (!(flags & GCWT_OUT_ON_DEVICE)) {
    // complex rows are float pairs: the same routine moves (and widens) them
    const size_t k = hp.out_elem_bytes / sizeof(float);
    hipError_t he = p->host_out.drain(dout, k * (size_t)row_len, rows, k * (size_t)n_out, out,
                                      (flags & GCWT_OUT_F64) != 0, p->stream);
    if (he != hipSuccess) {
      return hip_err(he, "host_out.drain");
    }
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->profiling) {
    // A stage's time is the length of the union of its spans (the level passes run on three
    // streams beside each other: their sum would count the same wall time up to three times).
    float acc[ST_COUNT] = {0};
    if (!p->spans.empty()) {
      const hipEvent_t t0 = p->spans.front().a;
      std::vector<std::pair<float, float>> iv[ST_COUNT];
      float last_end = 0;
      for (const auto& s : p->spans) {
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, t0, s.a));
        HIP_TRY(hipEventElapsedTime(&b, t0, s.b));
        iv[s.stage].push_back({a, b});
        if (s.stage == ST_INTERP) iv[ST_SYNTH].push_back({a, b});   // synth_ms: every synthesis kernel
        last_end = std::max(last_end, b);
      }
      for (int st = 0; st < ST_COUNT; ++st) {
        std::sort(iv[st].begin(), iv[st].end());
        float lo = 0, hi = -1;
        for (const auto& x : iv[st]) {
          if (hi < lo || x.first > hi) { if (hi >= lo) acc[st] += hi - lo; lo = x.first; hi = x.second; }
          else hi = std::max(hi, x.second);
        }
        if (hi >= lo) acc[st] += hi - lo;
      }
      p->last.total_ms = last_end;
    }
    p->last.mean_ms = acc[ST_MEAN];
    p->last.fwd_fft_ms = acc[ST_FWD];
    p->last.decimate_ms = acc[ST_DECIM];
    p->last.block_fft_ms = acc[ST_BLOCK];
    p->last.synth_ms = acc[ST_SYNTH];
    p->last.interp_ms = acc[ST_INTERP];
    p->last.direct_ms = acc[ST_DIRECT];
    p->last.fullband_ms = acc[ST_FULLBAND];
    p->last.blockconv_ms = acc[ST_BLOCKCONV];
    p->have_timings = true;
  }
  return GCWT_OK;
}

static int gcwt_execute_impl(gcwt_plan* p, const void* x, void* out, int flags) {
  if (!p || !x || !out) return set_err(GCWT_ERR_INVALID, "NULL argument");
  return execute_range(p, x, out, 0, p->hp.prm.n_samples, flags & ~GCWT_REUSE_MEANS);
}

static int gcwt_execute_block_impl(gcwt_plan* p, const void* x, void* out, int64_t start, int64_t length,
                       int flags) {
  if (!p || !x || !out) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (start < 0 || length <= 0 || start + length > p->hp.prm.n_samples)
    return set_err(GCWT_ERR_INVALID, "block range outside the recording");
  return execute_range(p, x, out, start, start + length, flags);
}

int gcwt_plan_segment_count(const gcwt_plan* p) { return p ? (int)p->hp.epochs.size() : -1; }

int gcwt_plan_segment_info(const gcwt_plan* p, int segment, int64_t* core_start, int64_t* core_stop,
                           int64_t* fft_length) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  if (segment < 0 || segment >= (int)p->hp.epochs.size())
    return set_err(GCWT_ERR_INVALID, "segment out of range");
  const EpochPlan& ep = p->hp.epochs[segment];
  if (core_start) *core_start = ep.core0;
  if (core_stop) *core_stop = ep.core1;
  if (fft_length) *fft_length = ep.p;
  return GCWT_OK;
}

int gcwt_filter_bank(gcwt_plan* p, float* bank) {
  if (!p || !bank) return set_err(GCWT_ERR_INVALID, "NULL argument");
  int rc = gcwt_plan_upload(p);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(bank, p->d_bank, sizeof(float2) * (size_t)p->hp.prm.n_freqs * p->hp.block,
                    hipMemcpyDeviceToHost));
  return GCWT_OK;
}

int gcwt_direct_kernel(gcwt_plan* p, int scale, float* psi) {
  if (!p || !psi) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (scale < 0 || scale >= p->hp.prm.n_freqs) return set_err(GCWT_ERR_INVALID, "scale out of range");
  const ScalePlan& s = p->hp.scales[scale];
  if (s.method != GCWT_SCALE_DIRECT) return set_err(GCWT_ERR_INVALID, "not a direct scale");
  int rc = gcwt_plan_upload(p);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(psi, p->d_psi_lit + s.direct_offset + direct_front_pad(s.length), sizeof(float2) * (size_t)s.length,
                    hipMemcpyDeviceToHost));
  return GCWT_OK;
}

int gcwt_plan_precision_report(const gcwt_plan* p, float* predicted, float* worst, int32_t* n_rerouted) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  const size_t S = (size_t)p->hp.prm.n_freqs;
  if (predicted)
    for (size_t i = 0; i < S; ++i) predicted[i] = i < p->last_pred.size() ? p->last_pred[i] : 0.f;
  if (worst) *worst = p->last_worst;
  if (n_rerouted) *n_rerouted = p->last_rerouted;
  return GCWT_OK;
}

// test hook: the two terms of the prediction (rounding of the level's stages; what the level leaves out) of workspace
// slot 0 of the last batch the last execute ran, S floats each, and the level energies (n_levels floats, may be NULL)
int gcwt_debug_precision_terms(gcwt_plan* p, float* rounding, float* left_out, float* level_energy, float* band_energy) {
  if (!p || !rounding || !left_out) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (!p->detect || p->last_batch_slots <= 0) return set_err(GCWT_ERR_INVALID, "no execute with the detector on yet");
  const int S = p->hp.prm.n_freqs, L = (int)p->hp.levels.size();
  // scratch predictions: the kernel writes pred[(segment, channel, scale)] -- every slot's segment is 0 here, so C rows of S
  struct Scratch {
    float* p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
  } sc, lv, pr;
  const size_t Cn = (size_t)p->hp.prm.n_channels;
  HIP_TRY(hipMalloc((void**)&sc.p, sizeof(float) * 2 * (size_t)S * p->last_batch_slots));
  HIP_TRY(hipMalloc((void**)&lv.p, sizeof(float) * (size_t)L * p->last_batch_slots));
  HIP_TRY(hipMalloc((void**)&pr.p, sizeof(float) * (size_t)S * Cn));
  PredSegs dbg_segs{};
  dbg_segs.n_channels = p->hp.prm.n_channels;
  hipError_t he = launch_precision_predict(p->d_bands, p->d_gain, p->d_scale_level, p->d_scale_length, p->ep_dev[p->last_batch].pred_levels, S, L,
                                           p->last_pt, p->kappa_eps, p->oob_tol, pr.p, lv.p, sc.p, p->last_batch_slots, dbg_segs, p->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(rounding, sc.p, sizeof(float) * (size_t)S, hipMemcpyDeviceToHost, p->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(left_out, sc.p + S, sizeof(float) * (size_t)S, hipMemcpyDeviceToHost, p->stream);
  if (he == hipSuccess && level_energy) he = hipMemcpyAsync(level_energy, lv.p, sizeof(float) * (size_t)L, hipMemcpyDeviceToHost, p->stream);
  if (he == hipSuccess && band_energy) he = hipMemcpyAsync(band_energy, p->d_bands, sizeof(float) * (size_t)kSpecBands, hipMemcpyDeviceToHost, p->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(p->stream);
  if (he != hipSuccess) return hip_err(he, "precision terms");
  return GCWT_OK;
}

int gcwt_get_timings(const gcwt_plan* p, gcwt_timings* t) {
  if (!p || !t) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (!p->have_timings) return set_err(GCWT_ERR_INVALID, "no profiled execute yet");
  *t = p->last;
  return GCWT_OK;
}

// ---- exception-safe entry points ------------------------------------------
int gcwt_plan_create(gcwt_plan** out, const gcwt_params* params) {
  return guarded([&] { return gcwt_plan_create_impl(out, params); });
}

int gcwt_plan_upload(gcwt_plan* p) {
  return guarded([&] { return gcwt_plan_upload_impl(p); });
}

int gcwt_execute(gcwt_plan* p, const void* x, void* out, int flags) {
  return guarded([&] { return gcwt_execute_impl(p, x, out, flags); });
}

int gcwt_execute_block(gcwt_plan* p, const void* x, void* out, int64_t start, int64_t length,
                       int flags) {
  return guarded([&] { return gcwt_execute_block_impl(p, x, out, start, length, flags); });
}


// ---- debug hooks (include/ghostcwt_debug.h) --------------------------------
int gcwt_debug_measure_build(void) { return kMeasureBuild ? 1 : 0; }

int gcwt_debug_level_count(const gcwt_plan* p) { return p ? (int)p->hp.levels.size() : -1; }

int gcwt_debug_level_info(const gcwt_plan* p, int epoch, int level, int32_t* decimation,
                          int32_t* halo, int32_t* hop, int32_t* nblk, int64_t* m) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  if (epoch < 0 || epoch >= (int)p->hp.epochs.size() || level < 0 ||
      level >= (int)p->hp.levels.size())
    return set_err(GCWT_ERR_INVALID, "epoch/level out of range");
  const LevelPlan& lp = p->hp.levels[level];
  const EpochLevel& el = p->hp.epochs[epoch].lv[level];
  if (decimation) *decimation = lp.decimation;
  if (halo) *halo = lp.halo;
  if (hop) *hop = lp.hop;
  if (nblk) *nblk = el.nblk;
  if (m) *m = el.m;
  return GCWT_OK;
}

int gcwt_debug_level_band_shift(const gcwt_plan* p, int level, int32_t* shift) {
  if (!p || !shift) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (level < 0 || level >= (int)p->hp.levels.size()) return set_err(GCWT_ERR_INVALID, "level out of range");
  *shift = p->hp.levels[level].band_shift;
  return GCWT_OK;
}

int gcwt_debug_set_option(const char* name, int64_t value, int clear) {
  const int rc = option_set(name, (long long)value, clear != 0);
  if (rc == -1) return set_err(GCWT_ERR_INVALID, std::string("unknown option: ") + (name ? name : "(null)"));
  if (rc == -2) return set_err(GCWT_ERR_UNSUPPORTED, std::string("option ") + name + " exists in libghostcwt_measure.so only");
  return GCWT_OK;
}

int gcwt_debug_level_low_cut(const gcwt_plan* p, int level, double* theta_cut) {
  if (!p || !theta_cut) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (level < 0 || level >= (int)p->hp.levels.size()) return set_err(GCWT_ERR_INVALID, "level out of range");
  const LevelPlan& own = p->hp.levels[p->hp.levels[level].xr_owner];
  *theta_cut = p->hp.high_precision && own.band_shift == 0 ? own.taper_hi : 0.0;
  return GCWT_OK;
}

int gcwt_debug_scale_theta_lo(const gcwt_plan* p, double* theta_lo) {
  if (!p || !theta_lo) return set_err(GCWT_ERR_INVALID, "NULL argument");
  for (size_t i = 0; i < p->hp.scales.size(); ++i) theta_lo[i] = p->hp.scales[i].theta_lo;
  return GCWT_OK;
}

int gcwt_debug_mean_folded(const gcwt_plan* p) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  return p->last_fold ? 1 : 0;
}

int gcwt_debug_graph_state(const gcwt_plan* p) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  return p->graph_failed ? -1 : p->graph_exec ? 1 : 0;
}

int gcwt_debug_blockconv_groups(const gcwt_plan* p, int32_t* first, int32_t* count, int32_t* hop, int32_t* back,
                                int32_t* order, int max_groups, int max_order) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  const HostPlan& hp = p->hp;
  const int n = (int)hp.bc_groups.size();
  for (int g = 0; g < n && g < max_groups; ++g) {
    if (first) first[g] = hp.bc_groups[g].first;
    if (count) count[g] = hp.bc_groups[g].count;
    if (hop) hop[g] = hp.bc_groups[g].hop;
    if (back) back[g] = hp.bc_groups[g].back;
  }
  if (order) for (int k = 0; k < hp.n_blockconv && k < max_order; ++k) order[k] = hp.bc_order[k];
  return n;
}

int gcwt_debug_interp_level(const gcwt_plan* p, int level, int32_t* q, int32_t* factor, double* alpha,
                            double* err_bound, float* coef, int64_t max_floats) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  if (level < 0 || level >= (int)p->hp.levels.size()) return set_err(GCWT_ERR_INVALID, "level out of range");
  const LevelPlan& lp = p->hp.levels[level];
  const bool on = level_kernel(p, lp) == LK_INTERP;
  if (q) *q = on ? lp.interp_q : 0;
  if (factor) *factor = on ? lp.interp_factor : 0;
  if (alpha) *alpha = lp.interp_alpha;
  if (err_bound) *err_bound = lp.interp_err;
  if (coef && on) {
    const int64_t n = (int64_t)2 * lp.interp_factor * 8;
    if (n > max_floats) return set_err(GCWT_ERR_INVALID, "destination too small");
    memcpy(coef, p->hp.interp_coef.data() + lp.coef_offset, sizeof(float) * (size_t)n);
  }
  return GCWT_OK;
}

int gcwt_debug_scale_demod(const gcwt_plan* p, int32_t* demod) {
  if (!p || !demod) return set_err(GCWT_ERR_INVALID, "NULL argument");
  for (size_t i = 0; i < p->hp.scales.size(); ++i) demod[i] = p->hp.scales[i].demod_bin;
  return GCWT_OK;
}

int gcwt_debug_scale_theta_neg(const gcwt_plan* p, double* theta_neg) {
  if (!p || !theta_neg) return set_err(GCWT_ERR_INVALID, "NULL argument");
  for (size_t i = 0; i < p->hp.scales.size(); ++i) theta_neg[i] = p->hp.scales[i].theta_neg;
  return GCWT_OK;
}

int gcwt_debug_batch_of(const gcwt_plan* p, int segment, int32_t* first, int32_t* count) {
  if (!p) return set_err(GCWT_ERR_INVALID, "NULL plan");
  if (segment < 0 || segment >= (int)p->hp.epochs.size())
    return set_err(GCWT_ERR_INVALID, "segment out of range");
  const int f = p->hp.epochs[segment].batch_first;
  if (first) *first = f;
  if (count) *count = p->hp.epochs[f].batch_count;
  return GCWT_OK;
}

int gcwt_debug_exact_gain(const gcwt_plan* p, int scale, const int64_t* a, int64_t b, int64_t n,
                          double* gain) {
  if (!p || !a || !gain || b <= 0) return set_err(GCWT_ERR_INVALID, "bad argument");
  if (scale < 0 || scale >= (int)p->hp.scales.size()) return set_err(GCWT_ERR_INVALID, "scale out of range");
  const ScalePlan& s = p->hp.scales[scale];
  for (int64_t i = 0; i < n; ++i)
    gain[i] = exact_gain(p->hp.amps.data() + s.amp_offset, s.bin_lo, s.n_bins, s.length, a[i], b);
  return GCWT_OK;
}

int gcwt_debug_clock(gcwt_plan* p, double* ghz, double* workgroup_seconds) {
  if (!p || !ghz) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (!p->d_probe)
    return set_err(GCWT_ERR_INVALID, kMeasureBuild ? "plan was not created with GHOSTCWT_CLOCK_PROBE=1"
                                                   : "the clock probe exists only in libghostcwt_measure.so (make measure)");
  unsigned long long v[8] = {};
  HIP_TRY(hipMemcpy(v, p->d_probe, sizeof(v), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(p->d_probe, 0, sizeof(v)));
  *ghz = v[1] ? (double)v[0] / (double)v[1] * 0.1 : 0.0;
  if (workgroup_seconds) *workgroup_seconds = (double)v[1] * 1e-8;
  if (kMeasureBuild && option_is_set("clock_phases") && v[6])   // mean microseconds into a k_synth7 workgroup's life at its marks
    fprintf(stderr, "k_synth7 workgroups %llu: loads parked %.2f us, spectra exchanged %.2f, spectra done %.2f, loop starts %.2f, ends %.2f\n",
            v[6], v[2] * 0.01 / v[6], v[3] * 0.01 / v[6], v[4] * 0.01 / v[6], v[5] * 0.01 / v[6], v[1] * 0.01 / v[6]);
  return GCWT_OK;
}

int gcwt_debug_fetch(gcwt_plan* p, int what, int channel, int epoch, int level, float* dst,
                     int64_t max_complex) {
  if (!p || !dst) return set_err(GCWT_ERR_INVALID, "NULL argument");
  if (!p->uploaded) return set_err(GCWT_ERR_INVALID, "plan has not run");
  const HostPlan& hp = p->hp;
  if (epoch < 0 || epoch >= (int)hp.epochs.size() || channel < 0 || channel >= hp.prm.n_channels)
    return set_err(GCWT_ERR_INVALID, "epoch/channel out of range");
  const EpochPlan& ep = hp.epochs[epoch];
  // workspace slot of (segment, channel) after a full execute: position in its batch
  const int64_t slot = (int64_t)(epoch - ep.batch_first) * hp.prm.n_channels + channel;
  const float2* src = nullptr;
  int64_t n = 0;
  if (what == GCWT_DEBUG_SPECTRUM) {
    src = p->d_x + slot * ep.p_store;
    n = ep.p_store;
  } else {
    if (level < 0 || level >= (int)hp.levels.size()) return set_err(GCWT_ERR_INVALID, "level out of range");
    if (what == GCWT_DEBUG_DECIMATED) {
      src = p->d_xr + slot * hp.max_xr + ep.lv[level].xr_offset;
      n = ep.lv[level].m;
    } else if (what == GCWT_DEBUG_BLOCK_SPECTRA) {
      src = p->d_xb + slot * hp.max_xb + ep.lv[level].xb_offset;
      n = (int64_t)ep.lv[level].nblk * hp.block;
    } else {
      return set_err(GCWT_ERR_INVALID, "unknown buffer");
    }
  }
  if (n > max_complex) return set_err(GCWT_ERR_INVALID, "destination too small");
  HIP_TRY(hipMemcpy(dst, src, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost));
  return (int)0;
}

}  // extern "C"

// accessors for comm.cpp
int gcwt_internal_refresh_bank(gcwt_plan* p) {   // derived tables follow a (broadcast) bank
  hipError_t he = launch_bank_gain(p->d_bank, p->d_gain, p->d_bank_sc, p->hp.prm.n_freqs, p->stream);
  if (he != hipSuccess) return hip_err(he, "bank_gain");
  he = launch_scale_windows(p->d_gain, p->d_scale_list, p->n_listed, (float)p->hp.band_tol, p->d_gain_lv,
                            p->prune_inputs, p->stream);
  if (he != hipSuccess) return hip_err(he, "scale_windows");
  return GCWT_OK;
}
int gcwt_internal_set_error(int code, const char* msg) { return set_err(code, msg); }
float2* gcwt_internal_bank_ptr(gcwt_plan* p, size_t* bytes) {
  *bytes = sizeof(float2) * (size_t)p->hp.prm.n_freqs * p->hp.block;
  return p->d_bank;
}
hipStream_t gcwt_internal_stream(gcwt_plan* p) { return p->stream; }
int gcwt_internal_device(gcwt_plan* p) { return p->device; }
