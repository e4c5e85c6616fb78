// kernels.h -- device-side argument structs and launch wrappers (see kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/ghostcwt.h"

namespace gcwt {

constexpr int kRowLenDev = 4096;

// Measurement hooks (stores dropped, barriers removed, in-kernel clock probe, the slower
// k_synth8) exist only in the second build, `make measure` -> libghostcwt_measure.so, which
// tools/ load through GHOSTCWT_LIB; the product library has none of those paths.
#ifdef GCWT_MEASURE
constexpr bool kMeasureBuild = true;
#else
constexpr bool kMeasureBuild = false;
#endif

struct BankScale {
  double omega;
  double half_delay;
  int64_t length;      // L of the reference kernel (morse.py:108-122)
  int64_t amp_offset;  // kept spectrum samples A_j: amps[amp_offset .. + n_bins), bins bin_lo ..
  int32_t bin_lo, n_bins;
  int32_t decimation;
  int32_t spectral;
  int32_t band_shift;  // bank bin k is the frequency (k - band_shift) 2 pi / (256 R): the level's band starts
  int32_t pad;         // that many bins below zero (planner.h: LevelPlan::band_shift)
};

struct DirectScale {
  double omega;
  int64_t length;
  int64_t amp_offset;   // kept spectrum samples A_j: amps[amp_offset .. + n_bins), bins bin_lo ..
  int64_t offset;       // into the psi buffer
  int32_t scale;        // output row
  int32_t bin_lo, n_bins;
  int32_t front;        // zero taps in front of the kernel in the psi buffer (direct_front_pad)
};

// k_direct wants (L-1)/2 + front = 7 (mod 8): see the kernel's header
inline int direct_front_pad(int64_t length) { return (int)((7 - (length - 1) / 2) & 7); }

// Segments launched together (planner.h: batches).  blockIdx.y = segment * n_channels +
// channel; every workspace array simply has n_channels * n_segments "channels", only the
// kernels that touch the caller's arrays need the per-segment numbers below.
constexpr int kSegBatch = 16;
struct SegIn {                       // forward column pass: where the segment's samples are
  int64_t x_off[kSegBatch];          // offset of segment sample 0 inside a channel's row
  int64_t n_valid[kSegBatch];        // segment-local samples [n_lead, n_valid) exist, the
  int64_t n_lead[kSegBatch];         //   rest read as zero
  // precision = high, time blocks of a long epoch: the block's own edges (not the epoch's) are faded over ramp_lo /
  // ramp_hi samples beyond the halo its outputs reach (planner.h: EpochPlan::ramp_*); 0: a hard edge
  int64_t ramp_lo[kSegBatch], ramp_hi[kSegBatch];
  int32_t n_channels, pad;
};
struct SegOut {                      // synthesis: where the segment's results go
  int64_t seg_col[kSegBatch];        // column of segment sample 0 in an out row (may be < 0)
  int64_t w_lo[kSegBatch];           // segment-local samples [w_lo, w_hi) are written
  int64_t w_hi[kSegBatch];
  int32_t n_channels, pad;
};

struct SynthItemDev {
  int32_t level, scale, blk0, nblk;
};

struct SynthLevelDev {
  int32_t decimation, hop, halo, blk_base;   // block b of the level sits at XB index b - blk_base
  int64_t xb_offset;   // per-channel offset of this level's block spectra (complex elems)
  int64_t tw_offset;   // into level_tw
};

struct SynthArgs {
  const float2* xb;
  const float2* bank;
  const float2* tw256;
  const float2* level_tw;
  const SynthItemDev* items;
  const SynthLevelDev* levels;
  float* out;
  int64_t xb_cstride;
  int64_t row_len;       // samples per (channel, scale) row of out
  int32_t n_scales;
  int32_t pad;
  SegOut seg;
};

struct Synth7Item {
  int32_t level, blk0, rtile, pad;
};

struct Synth7Level {
  int32_t decimation, log2r, hop, halo, nblk, n_scales, scale_offset, blk_base;
  int32_t n_plain;     // the first n_plain scales of the level's list have odd L (real filter);
  int32_t half_offset; // the rest carry the half-sample phase level_half_tw[half_offset + k]
  int32_t band_shift;  // bins of the level's grid below zero frequency (planner.h; 0 for the default wavelet)
  int32_t pad1;
  int64_t xb_offset;   // per-channel offset of this level's block spectra (complex elems)
  int64_t tw_offset;   // into level_tw
  int64_t xr_offset;   // per-channel offset of this level's decimated signal x_R (complex elems)
  int64_t m_mask;      // M - 1, M = P / R samples of x_R (circular index)
};

struct Synth7Args {
  const float2* xb;
  const float2* bank;
  const float2* tw256;
  const float2* level_tw;
  const Synth7Item* items;
  const Synth7Level* levels;
  const int32_t* scale_list;   // scale indices grouped by level, odd kernel lengths first
  const float* gain;           // |H_s[k]|, [S][256]
  const float* gain_lv;        // the same rows in list order, [entry][t][16 j] (k_scale_windows)
  const float2* level_half_tw; // exp(-i pi k/(256 R)), 256 per level
  float* out;
  int64_t xb_cstride;
  int64_t row_len;       // samples per (channel, scale) row of out
  const float2* xr;      // non-NULL: the workgroup computes its blocks' spectra itself from x_R
  int64_t xr_cstride;    //   (k_synth7 only; the block-spectra pass and the XB array are skipped)
  float xb_scale;        //   1 / (256 P)
  int32_t pad0;
  unsigned long long* clock_probe;   // measure build only (GHOSTCWT_CLOCK_PROBE=1): [0] += shader cycles,
                                     // [1] += 100 MHz ticks each workgroup lived; NULL otherwise
  int32_t n_scales;
  int32_t drop_stores;   // measure build only (GHOSTCWT_SYNTH_DROP_STORES=bits; results are WRONG): 1 stores get
                         // an empty range (kernel time without HBM writes), 2 no workgroup barriers
  SegOut seg;
};

// wide_halo: the items' levels have block halos above 48 (k_synth7<.., WIDE>); the other launch takes the rest
hipError_t launch_synth7(int mode, int ncol, bool wide_halo, const Synth7Args& a, int n_items, int n_channels,
                         hipStream_t st);

// Interpolating synthesis (synthi.hip): one workgroup = one block of one level, all its scales.
struct SynthiItem {
  int32_t level, blk0;          // first block of the workgroup's group
  int32_t pass0, n_pass;        // the passes (groups of scale slots) of the level's walk it makes
  int32_t wt_lo, wt_hi;         // the wave-tasks (256 samples each) of every (block, scale) it interpolates and
                                // stores: [0, all) unless one pass is too much for a workgroup (wt_lo: multiple of 4)
};
struct SynthiLevel {
  int32_t decimation, q, log2q, factor;   // R, phases per (block, scale), I = R / q
  int32_t hop, halo, nblk, n_scales, scale_offset, blk_base;
  int32_t log2nb, taps; // 1 << log2nb consecutive blocks per workgroup; interpolator taps: 8, or 6 (the rows' middle six)
  int64_t tw_offset;    // into level_tw
  int64_t xr_offset;    // per-channel offset of this level's decimated signal x_R (complex elems)
  int64_t m_mask;       // M - 1, M = P / R samples of x_R (circular index)
  int64_t coef_offset;  // into coef: [2][I][kInterpTaps] floats
};
struct SynthiArgs {
  const float2* tw256;
  const float2* level_tw;
  const SynthiItem* items;
  const SynthiLevel* levels;
  const int32_t* scale_list;   // as Synth7Args: scale index | (16 - j_hi) << 24, grouped by level
  const int32_t* scale_aux;    // per list entry: demodulation bin | (kernel length is even) << 16
  const float* gain;           // G_s[k], [S][256]
  const float* coef;           // interpolator coefficients
  float* out;
  int64_t row_len;
  const float2* xr;
  int64_t xr_cstride;
  float xb_scale;              // 1 / (256 P)
  int32_t n_scales;
  int32_t channels_fastest;   // grid (slots, items) instead of (items, slots): see api.cpp, interp_channels_fastest
  SegOut seg;
};
hipError_t launch_synthi(int mode, const SynthiArgs& a, int n_items, int n_channels, hipStream_t st);
// Pipelined interpolating synthesis (synthp.hip): q = 2 levels with I = R / 2 <= 256; producer waves make the z of
// round n + 1 while consumer waves interpolate and store round n.
constexpr int kSynthpThreads = 512;
constexpr int kSynthpProducers = 2;      // waves
constexpr int kSynthpSlots = 4;          // z slots (block, scale) of a round: nb blocks x ns scales
constexpr int kSynthpMaxFactor = 256;    // I: a lane's sub-sample positions depend on the lane alone
struct SynthpItem {
  int32_t level, blk0;          // first block of the workgroup's group of 1 << log2nb
  int32_t round0, n_rounds;     // the rounds (4 >> log2nb scales each) of the level's walk it makes
};
struct SynthpLevel {
  int32_t decimation, factor;   // R, I = R / 2
  int32_t hop, halo, nblk, n_scales, scale_offset, blk_base;
  int32_t log2nb;               // 1 << log2nb consecutive blocks per workgroup (0, 1, 2)
  int32_t n_plain;              // the first n_plain scales of the level's list have odd L; the rest carry the
  int32_t half_offset;          //   half-sample phase level_half_tw[half_offset + k]
  int32_t help;                 // of 128: the share of a round's tasks the two producer waves take once their z is made
  int64_t tw_offset;            // into level_tw
  int64_t xr_offset;            // per-channel offset of this level's decimated signal x_R (complex elems)
  int64_t m_mask;               // M - 1, M = P / R samples of x_R (circular index)
  int64_t coef_offset;          // into coef: [2][I][T] floats (the first table, tau = rho / I, is the one used)
};
struct SynthpArgs {
  const float2* tw256;
  const float2* level_tw;
  const SynthpItem* items;
  const SynthpLevel* levels;
  const int32_t* scale_list;    // as Synth7Args
  const int32_t* scale_aux;     // per list entry: demodulation bin | (kernel length is even) << 16
  const float* gain_lv;         // gains in list order, [entry][t][16 j] (k_scale_windows)
  const float2* level_half_tw;  // exp(-i pi k/(256 R)), 256 per level
  const float* coef;
  float* out;
  int64_t row_len;
  const float2* xr;
  int64_t xr_cstride;
  float xb_scale;               // 1 / (256 P)
  int32_t n_scales;
  int32_t channels_fastest;
  int32_t flags;                // (unused)
  int32_t pad0;
  SegOut seg;
};
// exact0: the items' levels have I = 4 (a lane's first sample is z itself)
hipError_t launch_synthp(int mode, const SynthpArgs& a, int n_items, int n_channels, bool exact0, hipStream_t st);
// same work items and arguments as launch_synth7 (synth8.hip)
hipError_t launch_synth8(int mode, int ncol, const Synth7Args& a, int n_items, int n_channels,
                         hipStream_t st);

// the operator layer in float64 (ops64.hip): host in, host out; out holds (re, im) pairs
hipError_t dft_f64(const double* x, int64_t n, int is_complex, int inverse, double* out);
hipError_t fastconv_f64(const double* signal, int64_t n, int signal_is_complex, const double* kernel, int64_t m,
                        int kernel_is_complex, int64_t first, int64_t count, double* out);

// a rectangle of a device-resident result to the host (result_io.hip; include/ghostcwt.h: gcwt_rows_to_host)
hipError_t rows_to_host(const float* d_src, int64_t src_pitch, int64_t n_rows, int64_t row_elems, void* dst,
                        int64_t dst_pitch, bool widen, bool pinned);

// sums: channel_sum_doubles(n_channels) doubles -- the results, then the workgroups' partial
// sums (kernels.hip: k_channel_sum, k_channel_sum_final)
constexpr int kSumParts = 256;        // partial sums kept per channel (k_channel_sum itself cuts a row into kSumPartsOwn at most)
constexpr int kSumPartsOwn = 64;
constexpr size_t channel_sum_doubles(size_t n_channels) { return n_channels * (kSumParts + 1); }
hipError_t launch_channel_sum(const float* x, int64_t n, int n_channels, double* sums,
                              hipStream_t st);
hipError_t launch_build_bank(float2* bank, float* gain, const BankScale* sc, const double* amps,
                             int n_scales, int B, hipStream_t st);
// A level's scale list entry: scale index in the low 24 bits; in the top 8, 16 - j_hi: the
// synthesis may skip the first-pass inputs j >= j_hi (bins from 16 j_hi up) of that scale
// (k_scale_windows).
constexpr int kScaleIndexMask = 0x00FFFFFF;
hipError_t launch_scale_windows(const float* gain, int32_t* scale_list, int n_listed, float tol,
                                float* gain_lv, bool prune, hipStream_t st);
hipError_t launch_bank_gain(const float2* bank, float* gain, const BankScale* sc, int n_scales,
                            hipStream_t st);
// full-band scales (exact.hip): H[k] / P on the k1-major grid of a P-point spectrum, the
// product with a batch of spectra, and the crop / |.| / store of the inverse transform
hipError_t launch_fullband_filter(float2* h, const BankScale* sc, int scale, const double* amps,
                                  int p1, hipStream_t st);
hipError_t launch_fullband_mul(const float2* x, const float2* h, float2* z, int64_t p, int n_slots,
                               hipStream_t st);
// the fused passes: rows of X * H for up to four scales per pass over X (pass 1), columns + crop + |.| + store
// (pass 2; p1 = 256, 512, 1024)
constexpr int kFullbandSet = 4;
struct FullbandSet {
  const float2* h[kFullbandSet];     // responses, k1-major like one slot of x
  float2* z[kFullbandSet];           // [slots][P] each
  int32_t n, pad;
};
// twt[256 j + 16 t + a] = exp(+2 pi i (t + 16 j) a / 4096), as for launch_bc_scales
hipError_t launch_fullband_rows(const float2* x, const FullbandSet& set, int p1, int64_t x_cstride, int64_t z_cstride,
                                const float2* twt, const float2* tw256, int n_slots, hipStream_t st,
                                int group = 0);
bool fullband_cols_fused(int p1);
// P = 4 x 4096: both passes fused (kernels.hip: k_fullband4); up to kFullband4Scales scales per launch
constexpr int kFullband4Scales = 8;
hipError_t launch_fullband4(int mode, const float2* x, const float2* const* h, const int32_t* scales, int n, float* out,
                            int64_t x_cstride, const float2* twt, const float2* tw256, int n_scales, int64_t row_len,
                            const SegOut& seg, int n_slots, hipStream_t st);
hipError_t launch_fullband_cols(int mode, const float2* z, float* out, int p1, int64_t z_cstride,
                                const float2* tw4096, const float2* tw256, int scale, int n_scales,
                                int64_t row_len, const SegOut& seg, int n_slots, hipStream_t st);
hipError_t launch_fullband_store(int mode, const float2* y, float* out, int64_t p, int scale,
                                 int n_scales, int64_t row_len, const SegOut& seg, int n_segments,
                                 hipStream_t st);
// psi: per scale the running sums of its taps (tap L - 1 zero); tail[scale]: the sum of all taps (k_build_direct)
hipError_t launch_build_direct(float2* psi, const DirectScale* sc, int n_direct, int64_t max_len,
                               const double* amps, float2* tail, float2* psi_literal, hipStream_t st);
hipError_t launch_fft_cols(int sign, bool real_in, const void* in, float2* out, int len, int ld,
                           int64_t in_cstride, int64_t out_cstride, int64_t tw_n,
                           const float2* tw4096, const float2* tw256, const double* sums, double inv_n, int64_t n_valid,
                           int n_channels, hipStream_t st, int64_t n_lead = 0, int rows_out = 0);
// forward pass of a batch of segments: real input, grid.y = segs.n_channels * n_segments
hipError_t launch_fft_cols_batch(const float* in, float2* out, int len, int ld, int64_t in_cstride,
                                 int64_t out_cstride, int64_t tw_n, const float2* tw4096,
                                 const float2* tw256, const double* sums, double inv_n,
                                 const SegIn& segs, int n_segments, hipStream_t st, int rows_out);
// Low cut of a level's slice of the spectrum (precision = high; planner.h: LevelPlan::taper_hi): element
// i of row r is bin k = r + p1 i; it is multiplied by 0 for k <= k0, by 1 for k >= k0 + 1 / inv_width and
// by half a cosine in between.  p1 = 0: no cut.
struct RowTaper {
  int32_t p1 = 0;
  float k0 = 0.f, inv_width = 0.f;
};
hipError_t launch_fft_rows(int sign, const float2* in, float2* out, int len, int64_t n_rows,
                           int64_t in_ld, int64_t out_ld, int64_t in_cstride, int64_t out_cstride,
                           int64_t tw_n, const float2* tw4096, const float2* tw256, float scale,
                           int n_channels, hipStream_t st, int out_len = 0, int mirror = 0,
                           RowTaper taper = RowTaper());
// float64 forward transform (fwd64.hip): pass A real columns -> y (float64), pass B rows -> the float32
// k1-major spectrum; tables from fwd64_fill_tables (8192 double2)
void fwd64_fill_tables(double2* host);
hipError_t launch_fwd64_cols(const float* in, double2* y, int p1, int64_t in_cstride, int64_t y_cstride,
                             int64_t p, const double2* tables, const double* sums, double inv_n,
                             const SegIn& segs, int n_segments, int rows_out, hipStream_t st, int in_stride = 1,
                             int in_offset = 0, bool fold_mean = false);
// fold_mean (one segment that IS the recording, no faded edges, no interleaved transforms: api.cpp decides): the column pass transforms
// the samples as they are, leaves its workgroups' partial sums behind sums[C ..] (k_channel_sum's layout; added up by
// launch_channel_sum_final), and the row pass takes the mean's own transform out of Y as it reads it (fold_sums): the
// recording is read once for the forward side, not twice (transforms.py:142-143: x - mean(x)).
hipError_t launch_channel_sum_final(double* sums, int n_channels, int parts, hipStream_t st);
// workgroups per channel of the column pass (= partial sums it leaves): k_fwd64_cols256_real2 walks four 32-column
// tiles per workgroup, the others take 16 columns each
constexpr int fold_parts(int p1) { return p1 == 256 ? 32 : 256; }
// comb_n > 1 (long mode): subsequence comb_a of comb_n, accumulated into x with the twiddle W_p_true^(a k)
hipError_t launch_fwd64_rows(const double2* y, float2* x, int n_rows, int64_t y_cstride, int64_t x_cstride,
                             const double2* tables, int n_slots, int out_len, int mirror, hipStream_t st,
                             int comb_a = 0, int comb_n = 1, int64_t p_true = 0, float* hist = nullptr, int hist_rows = 0,
                             const double* fold_sums = nullptr, double inv_n = 0.0, int64_t n_valid = 0, int p1 = 0);

// precision = auto (detect.hip): what the float32 stages of the decimated path will cost each scale, predicted from
// the float64 spectrum while it is made.  The positive half of the spectrum is summed into bands, sixteen per octave:
// bin k >= 1 belongs to band (bits of (float) k >> 19) - 127 * 16, i.e. [2^e (1 + m / 16), 2^e (1 + (m + 1) / 16)).
constexpr int kSpecBands = 16 * 24;
__host__ __device__ inline int spec_band(float k) {
  union { float f; uint32_t u; } b;
  b.f = k;
  const int v = (int)(b.u >> 19) - 127 * 16;
  return v < 0 ? 0 : (v >= kSpecBands ? kSpecBands - 1 : v);
}
struct PredSegs {           // slot = g * n_channels + channel belongs to the plan's segment seg[g]: pred is [segments][C][S]
  int32_t seg[kSegBatch];
  int32_t n_channels, pad;
};
struct PredLevel {
  int32_t decimation;       // R: the level's x_R holds the spectrum bins below P / R (after its low cut)
  int32_t band_shift;       // bins of the level's 256-point grid below zero frequency (shifted bands: no low cut)
  float k0, k1;             // low cut in spectrum bins: zero below k0, raised to one at k1 (k1 <= k0: none)
};
// One workgroup per (slot, level): E_level = the band energies the level's x_R contains, E_s = those a scale's gains let
// through, W_s = sum G_s^2 / 256 (its share of a white noise floor); pred[s] = max over slots of
// kappa_eps sqrt(E_level W_s / E_s) -- the float32 rounding of the level transform and block spectra, white at
// ~2^-24 of the level's content, against the scale's own output -- and oob_tol sqrt(E_out / E_s), E_out what the
// level's x_R leaves out (the reference's kernel answers to it through its side lobes; the decimated path does
// not).  scale_level[s] < 0: not on the decimated path.  dbg_scale: [slots][2][S], the two terms.
// hist: [slots][p1][kRowBands], the energy |X|^2 of the bins 0 <= k < P / 2 of the k1-major spectrum ([slot][p1 rows][4096],
// bin k1 + p1 k2 at (k1, k2)) per (slot, row k1): entries [0, 16) the bins k2 < 16 one by one, [16, 128) the bands of
// k2 = 16 .. 2047 (band of k = band of k2 + 16 log2 p1 there) -- written by the forward row pass itself (fwd64.hip:
// row_band_sums; plain stores, every entry every time).  launch_band_sums adds the rows up, in row order, into
// bands: [slots][kSpecBands] (bin 0 left out).
constexpr int kRowBands = 128;
hipError_t launch_band_sums(const float* hist, int p1, float* bands, int n_slots, hipStream_t st);
hipError_t launch_precision_predict(const float* bands, const float* gain, const int32_t* scale_level,
                                    const int32_t* scale_length, const PredLevel* levels, int n_scales, int n_levels, double p_true,
                                    float kappa_eps, float oob_tol, float* pred, float* dbg_level, float* dbg_scale,
                                    int n_slots, const PredSegs& segs, hipStream_t st);
// shifted band of a level: Xs[k1][j2] = X[k1 + P1 (j2 - u2)] from the positive half of a real signal's
// k1-major spectrum (kernels.hip: k_shift_gather)
hipError_t launch_shift_gather(const float2* x, float2* xs, int p1, int q, int u2, int64_t x_row,
                               int64_t x_cstride, int64_t xs_cstride, int n_channels, hipStream_t st);
hipError_t launch_block_fft(const float2* xr, float2* xb, int64_t m, int hop, int halo, int blk_lo,
                            int nblk,
                            int64_t xr_cstride, int64_t xb_cstride, const float2* tw256, float scale,
                            int n_channels, hipStream_t st);
hipError_t launch_synth(int mode, const SynthArgs& a, int n_items, int n_channels, hipStream_t st);
// epoch [epoch_start, epoch_start + epoch_len); samples [g_lo, g_hi) of it are computed and
// written at column (n - col0) of rows of row_len samples
struct DirectEpochs {                // time-domain scales: epochs handled by one launch
  int64_t epoch_start[kSegBatch], epoch_len[kSegBatch];
  int64_t g_lo[kSegBatch], g_hi[kSegBatch];   // samples of the recording to produce
  int32_t n_channels, pad;
};
hipError_t launch_direct(int mode, const float* x, float* out, const float2* psi,
                         const DirectScale* sc, int n_direct, const double* sums, double inv_n,
                         int64_t n_samples, int n_scales, const DirectEpochs& eps, int n_epochs,
                         int64_t col0, int64_t row_len, int64_t max_len, const float2* tail, hipStream_t st,
                         const unsigned char* mask = nullptr);   // mask[scale row]: 0 = leave the row alone
// Block convolution (overlap-save; kernels.hip: k_bc_scales and fwd64.hip: k_bc_forward describe the path): the blocks of up to kSegBatch epochs that
// one launch handles.  Blocks are `hop` samples long and aligned to multiples of `hop` in recording time; block
// q of an epoch produces samples [q hop, (q + 1) hop) cut to [g_lo, g_hi) from the 4096 recording samples that
// start at q hop - back (those outside [epoch_start, epoch_stop) read as zero).  An epoch's blocks run from
// (g_lo / hop) & ~1 to ((g_hi - 1) / hop) | 1: whole even-aligned pairs, as the forward transform takes them
// whatever range is asked for (execute_block must give execute's bits).
struct BcBlocks {
  int64_t epoch_start[kSegBatch], epoch_stop[kSegBatch];
  int64_t g_lo[kSegBatch], g_hi[kSegBatch];     // samples of the recording to produce
  int32_t blk_first[kSegBatch + 1];             // blocks of epoch e: [blk_first[e], blk_first[e + 1])
  int32_t n_channels, n_epochs;
  int32_t hop, back;
  int32_t ramp, pad;                            // the block's first and last `ramp` samples fade in / out (smootherstep;
};                                              //   they lie outside what its outputs read: planner.h, BcGroup)
// spectra of the blocks [blk0, blk0 + nblk) of every channel: float64 transform of x - mean, rounded per bin;
// xb[(blk - blk0) * n_channels + ch][4096]
hipError_t launch_bc_forward(const float* x, float2* xb, const BcBlocks& bl, int blk0, int nblk,
                             int64_t n_samples, const double2* tables, const double* sums, double inv_n,
                             hipStream_t st);
// every scale of a group from those spectra: h[s][4096] the responses (k_fullband_filter with p1 = 1),
// rows[s] the output rows, twt[256 j + 16 t + a] = exp(+2 pi i (t + 16 j) a / 4096)
hipError_t launch_bc_scales(int mode, const float2* xb, float* out, const float2* h, const int32_t* rows,
                            int n_group_scales, const float2* twt, const float2* tw256, const BcBlocks& bl,
                            int blk0, int nblk, int n_scales, int64_t col0, int64_t row_len, hipStream_t st,
                            const unsigned char* mask = nullptr);
hipError_t launch_level_small(const float2* x, float2* xr, int n1, int q, int64_t p1_stride,
                              int64_t x_cstride, int64_t xr_cstride, const float2* tw4096,
                              int n_channels, hipStream_t st, RowTaper taper = RowTaper());
hipError_t launch_cmul_inplace(float2* a, const float2* b, int64_t n, hipStream_t st);
// Bluestein pieces (arbitrary-length DFT, analytic signal)
hipError_t launch_chirp_kernel(float2* b, int64_t N, int64_t P, hipStream_t st);
hipError_t launch_chirp_load(const float* v, int is_complex, int conj_in, int64_t n_valid, int64_t N,
                             int64_t P, const double* sums, double inv_n, float2* a, hipStream_t st);
hipError_t launch_chirp_analytic_mask(float2* y, int64_t N, int64_t P, float scale, hipStream_t st);
hipError_t launch_chirp_store(const float2* y, float2* out, int64_t count, int64_t N, float scale,
                              int conj_out, const double* sums, double inv_n, hipStream_t st);
hipError_t launch_zero_range(float* out, int64_t row_len_floats, int64_t n_rows, int64_t start,
                             int64_t len, hipStream_t st);

}  // namespace gcwt
