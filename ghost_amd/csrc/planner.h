// planner.h -- host-side planning for the CWT engine (no device code).
//
// Decides, for every analysis frequency, how it is evaluated -- the exact response of
// the reference's L-tap kernel on a decimated block grid, the literal kernel in the time
// domain, or a full-band FFT convolution -- from the MEASURED support of that kernel in
// frequency and in time (morse_exact.h), and lays out the per-epoch FFT sizes,
// decimation levels and block grids.
// Mirrors the set-up part of ghost/wave/transforms.py:179-185 and the length
// rule of ghost/wave/morse.py:108-122; the block/decimation layout is this
// engine's own.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ghostcwt.h"

namespace gcwt {

constexpr int kRowLen = 4096;      // row length of the two-pass big FFT (P = P1 * 4096)
constexpr int kMaxP1 = 1024;       // a stored spectrum has at most 2^22 bins (longer FFTs: EpochPlan::long_a)
constexpr int kMaxFftLog2 = 24;    // longest FFT of a segment; above 2^22 in "long mode" (EpochPlan::long_a)
constexpr int kMaxDecimation = 16384;
constexpr int kMaxBatch = 16;                // segments per launch set (kernels.h: kSegBatch)
constexpr int kMaxTwoPassDecimation = 256;   // above it the level IFFT uses the small-size kernel
constexpr int kSynthCols = 16;     // columns (block, r) per batch of the 16-column kernel
constexpr int kSynthWide = 32;      // columns per batch of the production kernel
constexpr int kDirectMaxLen = 256;  // longest kernel the time-domain path can take (k_direct's tile)
constexpr int kDirectDefaultLen = 48;   // ... and takes by default: ~6.4 us per tap and scale at 128 ch x 1e6, against
                                        // the block convolution's flat cost per scale
constexpr int kBlockConvRamp = 256;     // precision = exact: faded samples at either end of a block ...
constexpr int kBlockConvExactMaxLen = 1024;   // ... and the longest kernel that goes by blocks there: longer ones sit at
                                        // low frequencies, a few bins of a block from any drift, where 256 faded samples
                                        // do not keep its leakage out; the full-band path has no block edges
constexpr int kBlockConvMaxLen = 2560;  // longest kernel of the block convolution: its 4096-sample blocks then still
                                        // yield 1537 samples each; beyond it the full-band path (one FFT of the
                                        // whole segment per scale)

struct ScalePlan {
  double freq_hz = 0, omega = 0;   // omega = f / (fs/2) * pi   (transforms.py:408-410)
  int64_t length = 0;              // L = ceil(w0/omega * base) (morse.py:108-122)
  int method = GCWT_SCALE_SPECTRAL;
  int decimation = 1;              // R
  int level = -1;                  // index into HostPlan::levels
  double half_delay = 0;           // d = (L-1)/2 - (L-1)//2
  int direct_index = -1;           // index among direct scales
  int64_t direct_offset = 0;       // offset of psi in the direct-kernel buffer (complex elems)
  int fullband_index = -1;         // index among full-band scales
  int blockconv_index = -1;        // position in HostPlan::bc_order
  // kept spectrum samples A_j of the reference kernel that are not negligible (1e-18 of
  // the largest): bins bin_lo .. bin_lo + n_bins - 1, values HostPlan::amps[amp_offset ..]
  int32_t bin_lo = 0, n_bins = 0;
  int64_t amp_offset = 0;
  // measured on the exact response (planner.cpp: analyse_scale)
  double theta_hi = 0;             // |G| <= band_tol * peak for theta in [theta_hi, 2 pi - theta_neg]
  double theta_neg = 0;            // ... : how far below zero frequency the response still matters (the side
                                   // lobes of the L-tap truncation; 0 for the default wavelet)
  double theta_lo = 0;             // |G| <= low_tol * peak for theta in [0, theta_lo]: what a level may cut out of its
                                   // slice of the spectrum before the float32 stages see it (precision = high)
  double support = 0;              // samples either side of the centre that hold all but
                                   // support_tol of the kernel's energy (L2)
  bool band_ok = false;            // theta_hi + theta_neg <= pi: some decimation R >= 2 is exact to band_tol
  int demod_bin = 0;               // interpolated levels: bin of the level's 256-point grid the scale's
                                   // oversampled output is demodulated by (its band centre; a multiple of q)
};

struct LevelPlan {
  int decimation = 1;              // R
  int halo = 0;                    // Lh, decimated samples discarded at each block edge
  int hop = 0;                     // B - 2*Lh valid decimated samples per block
  std::vector<int> scales;         // scale indices evaluated on this level
  bool fast = true;                // 16 <= halo <= 48 and at most 256 scales: the production kernel
  int xr_owner = -1;               // first level with this decimation: its x_R is shared (a
                                   // decimation's scales are split by halo into up to two levels)
  int64_t twiddle_offset = 0;      // offset into the level twiddle table (complex elems)
  int band_shift = 0;              // bins of the level's 256-point grid that lie BELOW zero frequency: the level's
                                   // band is [-band_shift, 256 - band_shift) * 2 pi / (256 R); x_R is made from that
                                   // slice of the spectrum (heavy-tailed wavelets; 0 for the default one)
  double taper_hi = 0;             // precision = high, set on the level that owns x_R: the slice of the spectrum x_R is made
                                   // from is cut to zero below taper_hi / 2 and raised (half a cosine) to one at taper_hi
                                   // rad / sample -- below every gain of every scale that reads this x_R; 0: no cut
  // Interpolating synthesis (synthi.hip, interp.h; amplitude and power only): the level's scales
  // are made at q x the level's rate and brought to the full rate by a T-tap polyphase FIR.
  int interp_q = 0;                // phases of the 256-point inverse FFT per (block, scale); 0: not interpolated
  int interp_taps = 8;             // taps of the interpolator: 8, or 6 (the middle six of the table's rows)
  int interp_factor = 0;           // I = R / q
  int64_t coef_offset = 0;         // into HostPlan::interp_coef: [2][I][T] floats (odd / even kernel lengths)
  double interp_alpha = 0;         // design band of the interpolator, fraction of the oversampled Nyquist
  double interp_err = 0;           // bound on what the interpolation adds, relative to a scale's peak gain
};

struct EpochLevel {
  int64_t m = 0;                   // decimated length M = P / R
  int blk_lo = 0;                  // first block that touches the segment's output range
  int nblk = 0;                    // blocks blk_lo .. blk_lo + nblk - 1 are computed
  int64_t xr_offset = 0;           // per-channel offsets, complex elements
  int64_t xb_offset = 0;
};

struct SynthItem {                 // one workgroup of the synthesis kernel
  int32_t level, scale, blk0, nblk;
};

// One FFT-sized piece of work: a whole epoch, or a time block of a long epoch with a
// halo of input on each side.  Input samples [start, stop) (zero outside the epoch);
// `start` is a multiple of 64 so that every block's output lands on whole 128-byte lines
// of rows that are themselves aligned; output samples [core0, core1).
struct EpochPlan {
  int64_t start = 0, stop = 0, ne = 0;
  int64_t core0 = 0, core1 = 0;
  int64_t lead = 0;                // leading samples of the segment that lie before the epoch (zero)
  // Time blocks, precision = high: a block cut out of a long epoch ends where the recording does not, and a hard cut
  // leaks whatever the recording carries at low frequencies into every bin of the block's spectrum (1/k) -- exact
  // arithmetic cancels it in the block's core, the float32 level and block stages do not: 2.9e-4 on a 1/f^3 recording
  // at config 5's geometry (round 4).  The block's input is therefore extended beyond the halo its outputs reach by a
  // ramp on each side that is not an edge of the epoch, and faded to zero there (C2 "smootherstep"): ramp_lo / ramp_hi
  // segment samples at its start / end carry the weight; the core's outputs never see them.
  int64_t ramp_lo = 0, ramp_hi = 0;
  int epoch = 0;
  int64_t p = 0;                   // FFT length of this epoch
  int p1 = 0;                      // p_store = p1 * kRowLen rows of the stored spectrum
  // Long mode (round 4): a segment whose FFT is longer than 2^22 (a kernel of millions of taps: below 0.13 Hz at
  // 30 kHz) is transformed as long_a = p / 2^22 interleaved subsequences x[A n + a], each through the two-pass
  // FFT of p_store = p / A points, combined bin by bin: X[k] = sum_a W_p^(a k) X_a[k] for k < p_store / 2 -- the
  // only bins a plan reads whose levels all have R >= 2 A.  Everything after the forward transform sees a
  // spectrum of p_store bins in the usual k1-major layout and levels of decimation R / A of it.
  int long_a = 1;
  int64_t p_store = 0;
  std::vector<EpochLevel> lv;      // one per HostPlan::levels; shared by the segment's batch
  std::vector<SynthItem> items;
  int64_t xr_total = 0, xb_total = 0;  // per-channel complex elements
  // Segments of equal FFT length are launched together, each as an extra set of "channels"
  // (many short epochs, or the time blocks of a long one): batch_first is the first segment
  // of this one's batch, batch_count (set on that first segment) how many follow it.
  int batch_first = 0, batch_count = 1;
};

struct HostPlan {
  gcwt_params prm{};
  std::vector<double> freqs;
  std::vector<int64_t> bounds;
  int block = 256;                 // B
  double band_tol = 2e-7;          // out-of-band response tolerated, relative to the peak
  double support_tol = 4.5e-6;     // kernel energy (L2, relative) a block halo may cut off
  double low_tol = 2e-8;           // response below a scale's band a level's low cut may drop, relative to the peak
                                   // (the L-tap truncation's side lobes sit at 5e-9 .. 1e-8 there for the default wavelet)
  bool high_precision = true;      // gcwt_params.precision: float64 forward transform + per-level low cut
  double w0 = 0;                   // (beta/gamma)^(1/gamma)          (morseutils.py:315)
  double base_length = 0;          // 2 sqrt2 sqrt(gamma beta)/w0 * 4 (morse.py:115-116)
  double u_lo = 0, u_hi = 0;       // continuous spectrum above 1e-18 of its peak on [u_lo, u_hi] * omega
  std::vector<double> amps;        // A_j of every scale, back to back
  std::vector<ScalePlan> scales;
  std::vector<LevelPlan> levels;
  std::vector<EpochPlan> epochs;   // segments, in time order
  bool halo_static = true;         // every level is `fast`
  int n_direct = 0;
  int n_blockconv = 0;
  bool exact_only = false;         // precision = exact: no scale takes a decimated band or the time domain
  bool auto_precision = false;     // precision = default / auto: scales whose predicted float32 loss is too large are
                                   // made again by the exact paths (api.cpp); high predicts and reports only
  int direct_max_len = kDirectDefaultLen, blockconv_max_len = kBlockConvMaxLen;   // (options direct_max_len, blockconv)
  std::vector<int> bc_order;       // block-convolution scales by kernel length
  struct BcGroup {                 // consecutive entries of bc_order that share one set of block spectra
    int first = 0, count = 0;
    int hop = 0, back = 0;         // kernels.h: BcBlocks
    int ramp = 0;                  // precision = exact: samples at either end of a block that fade and that no output reads
  };
  std::vector<BcGroup> bc_groups;
  double bc_fill = 1.0;            // share of the block convolution's blocks that epochs fill (widest hop)
  int64_t bc_chunk_blocks = 0;     // blocks whose spectra the workspace holds at once (all channels)
  int n_fullband = 0;
  int max_bins = 0;                // largest n_bins
  int64_t direct_total = 0;        // complex elements of all direct kernels
  int64_t level_twiddle_total = 0;
  int64_t max_p = 0, max_xr = 0, max_xb = 0;
  int64_t max_p_store = 0;         // largest stored spectrum (= max_p unless a segment is in long mode)
  std::vector<float> interp_coef;  // interpolator coefficients of every interpolated level
  double interp_tol = 2e-7;        // largest interp_err a level may have and still be interpolated
  int max_fft_log2 = 22;
  int max_batch = 1;               // largest batch_count in the plan: workspace slots per channel
  size_t out_elem_bytes = 4;
  int64_t workspace_bytes = 0;
};

// Returns GCWT_OK or a negative status with *err filled.
int build_host_plan(const gcwt_params& prm, HostPlan* plan, std::string* err);

// Relative gain of the Morse filter at u times its peak frequency, as a log:
// beta ln u - (beta/gamma)(u^gamma - 1).
double morse_log_gain(double u, double gamma, double beta);

}  // namespace gcwt
