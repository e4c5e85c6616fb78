/* ghostcwt_debug.h -- test-only hooks of libghostcwt.so: read back the
 * intermediate buffers of the last gcwt_execute so each stage can be checked
 * against the NumPy model in tests/decimated_model.py.  Not part of the drop-in
 * surface. */
#ifndef GHOSTCWT_DEBUG_H
#define GHOSTCWT_DEBUG_H
#include "ghostcwt.h"
#ifdef __cplusplus
extern "C" {
#endif

enum {
  GCWT_DEBUG_SPECTRUM = 0,      /* X~[k1][k2] = X[k1 + P1 k2], P complex values; only
                                   k2 < 2048 (positive frequencies) is filled            */
  GCWT_DEBUG_DECIMATED = 1,     /* x_R of one level, M = P/R complex values (times P)    */
  GCWT_DEBUG_BLOCK_SPECTRA = 2  /* XB[blk][k] of one level, nblk*B complex values        */
};

/* Named switches read when a plan is created (csrc/options.h): which kernel instantiation or planning
 * layout is used -- "synth16", "synth_cols", "fuse_blocks", "slow_fft", "level_streams", "interp",
 * "interp_grid", "interp_lgnb", "synth_streams", "merge_levels", "split_levels" (all compute the same
 * rows; the tests compare them) and the budgets "batch_bytes", "stage_floats".  Options that change
 * accuracy or exist for measurements ("halo_margin", "interp_q", "interp_min_r", "prune_inputs",
 * "clock_phases", "synth_kernel", "synth_drop_stores", "clock_probe", "synthi_pad_kb") are refused with
 * GCWT_ERR_UNSUPPORTED by the product library: libghostcwt_measure.so takes them.  clear != 0 returns
 * the option to its default.  Process-wide; not part of the drop-in surface. */
int gcwt_debug_set_option(const char* name, int64_t value, int clear);
int gcwt_debug_level_count(const gcwt_plan* plan);
int gcwt_debug_level_info(const gcwt_plan* plan, int epoch, int level, int32_t* decimation,
                          int32_t* halo, int32_t* hop, int32_t* nblk, int64_t* m);
/* Bins of a level's 256-point grid that lie below zero frequency (heavy-tailed wavelets: the level's
 * band is [-shift, 256 - shift) * 2 pi / (256 R); 0 for the default wavelet). */
int gcwt_debug_level_band_shift(const gcwt_plan* plan, int level, int32_t* shift);
/* precision = high: x_R of the level's decimation is made from a slice of the spectrum that is zero
 * below theta_cut / 2 and raised (half a cosine) to one at theta_cut radians per sample -- below the band
 * of every scale that reads it (planner.h: LevelPlan::taper_hi); 0: no cut.  Per scale: theta_lo, up to
 * which its exact response stays under 2e-8 of its peak. */
int gcwt_debug_level_low_cut(const gcwt_plan* plan, int level, double* theta_cut);
int gcwt_debug_scale_theta_lo(const gcwt_plan* plan, double* theta_lo);
/* Small device-resident executes replay a HIP graph from the third call with the same arguments on: 1 = a graph is
 * instantiated, 0 = none (yet), -1 = capture failed once and the plan runs eagerly. */
int gcwt_debug_graph_state(const gcwt_plan* plan);
/* 1 when the plan's last run took the channel sums inside the forward column pass and the mean's transform out of the
 * row pass's input (one segment that is the whole recording, FFTs up to 2^22 points, every scale spectral; option
 * fold_mean = 0 turns it off), 0 when a pass of its own over x made them (transforms.py:142-143 either way). */
int gcwt_debug_mean_folded(const gcwt_plan* plan);
/* precision = auto / high: the two terms of the last execute's prediction for workspace slot 0 of its last batch
 * (S floats each: float32 rounding of the level's stages; what the level's slice of the spectrum leaves out), the
 * energy each level's x_R held (one float per level; may be NULL) and the band energies they come from (384 floats,
 * may be NULL: sum of |X[k]|^2 over the bins 0 < k < P / 2 of band (bits of (float) k >> 19) - 127 * 16, sixteen
 * bands per octave -- summed by the forward row pass, csrc/fwd64.hip: row_band_sums). */
int gcwt_debug_precision_terms(gcwt_plan* plan, float* rounding, float* left_out, float* level_energy,
                               float* band_energy);
/* Block convolution (GCWT_SCALE_BLOCKCONV): the scales in order of kernel length (`order`: n_blockconv entries, at
 * most max_order are written) and the groups of consecutive entries that share the spectra of their blocks (at
 * most max_groups are written): blocks of `hop` output samples from the 4096 recording samples that start `back`
 * before them.  Returns the number of groups. */
int gcwt_debug_blockconv_groups(const gcwt_plan* plan, int32_t* first, int32_t* count, int32_t* hop,
                                int32_t* back, int32_t* order, int max_groups, int max_order);
/* Segments of equal FFT length are launched together: first segment and size of the batch
 * that `segment` belongs to. */
int gcwt_debug_batch_of(const gcwt_plan* plan, int segment, int32_t* first, int32_t* count);
int gcwt_debug_fetch(gcwt_plan* plan, int what, int channel, int epoch, int level, float* dst,
                     int64_t max_complex);
/* The exact (real) response G of one scale's reference kernel at theta = 2 pi a[i] / b,
 * i < n, evaluated on the HOST by the same code the device bank builder runs
 * (csrc/morse_exact.h).  Needs no GPU. */
int gcwt_debug_exact_gain(const gcwt_plan* plan, int scale, const int64_t* a, int64_t b, int64_t n,
                          double* gain);
/* Interpolating synthesis (csrc/synthi.hip) of one level: phases q of the block transform per
 * (block, scale) -- 0 when the level is made by the FFT-per-sample kernels -- the interpolation
 * factor I = R / q, the design band (fraction of the oversampled Nyquist) and the planner's bound
 * on the error the interpolation adds, relative to a scale's peak gain.  coef (may be NULL):
 * [2][I][8] floats, the 8-tap interpolators of the I sub-sample positions for kernels of odd and
 * of even length.  demod (may be NULL): per SCALE of the plan, the bin of its level's 256-point
 * grid its oversampled output is demodulated by.  Host only. */
int gcwt_debug_interp_level(const gcwt_plan* plan, int level, int32_t* q, int32_t* factor,
                            double* alpha, double* err_bound, float* coef, int64_t max_floats);
int gcwt_debug_scale_demod(const gcwt_plan* plan, int32_t* demod);
/* Per scale: how far below zero frequency (radians per sample) the response of its kernel still
 * exceeds band_eps of the peak (planner.h: ScalePlan::theta_neg; 0 for the default wavelet). */
int gcwt_debug_scale_theta_neg(const gcwt_plan* plan, double* theta_neg);
/* 1 in libghostcwt_measure.so (`make -C ghost_amd/csrc measure`: -DGCWT_MEASURE), the build that
 * carries the measurement hooks -- stores dropped / barriers removed (GHOSTCWT_SYNTH_DROP_STORES,
 * results WRONG), the in-kernel clock probe, the slower k_synth8 (GHOSTCWT_SYNTH_KERNEL=8);
 * 0 in the product library, which has none of those paths. */
int gcwt_debug_measure_build(void);
/* Measure build only, plans created with GHOSTCWT_CLOCK_PROBE=1 in the environment: the shader
 * clock the synthesis workgroups ran at since the last call (sum of s_memtime deltas over
 * sum of s_memrealtime deltas, one pair per workgroup: MI355X_MICROARCH.md "DVFS
 * give-back" item 6) and the summed workgroup lifetimes. */
int gcwt_debug_clock(gcwt_plan* plan, double* ghz, double* workgroup_seconds);
/* Measurement only: GB/s this device reaches on `bytes` (>= 64 MiB) of HBM with a plain
 * 16-byte fill, a 16-byte copy (read + write counted), or the store pattern of the
 * synthesis kernel (128-byte runs into 100 rows a megasample apart).  Best of three. */
enum { GCWT_BW_FILL = 0, GCWT_BW_COPY = 1, GCWT_BW_SYNTH_STORES = 2,
       GCWT_BW_SYNTHI_STORES = 3 /* k_synthi's: 1 KB runs per wave, 53 KB visits, four rows per pass, three
                                    256-thread workgroups per CU (profiles/r04_store_study.md) */ };
int gcwt_debug_bandwidth(int pattern, size_t bytes, double* gb_per_s);

/* Whole-result properties of a device-resident result of rows_per_channel x n_channels rows (at most 65535) of
 * n_valid_floats floats, row_pitch_floats apart: how many values are Inf / NaN, and how many values of a
 * channel c >= distinct differ in any bit from the same row of channel c % distinct (a caller that tiled
 * `distinct` recordings over its channels expects 0: bench.py). */
int gcwt_debug_check_output(const void* out_device, int64_t row_pitch_floats, int64_t n_valid_floats,
                            int32_t rows_per_channel, int32_t n_channels, int32_t distinct,
                            int64_t* n_nonfinite, int64_t* n_mismatched);

#ifdef __cplusplus
}
#endif
#endif
