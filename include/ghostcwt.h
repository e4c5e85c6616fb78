/* ghostcwt.h -- C ABI of libghostcwt.so, the MI355X (gfx950) engine behind
 * ghost_amd.wave.ContinuousWaveletTransform.transform().
 *
 * The reference (nelpy/ghost) has no FFI of its own: it is pure Python.  The
 * seam this library replaces is the body of
 *   ghost/wave/transforms.py:142-231   (mean removal, per-scale loop, abs)
 * including everything that loop calls:
 *   ghost/wave/morse.py:84-91, ghost/wave/morseutils.py:93-198   (Morse kernel)
 *   ghost/sigtools/convolution.py:16-87 / 89-216                 (FFT convolution)
 * Each entry point below names the reference lines whose work it does.  The
 * Python class in ghost_amd/wave/transforms.py binds these with ctypes;
 * INTEGRATION.md shows the stub a ghost maintainer would add.
 *
 * Conventions: plain C types only; every function returns 0 on success or a
 * negative gcwt_status; nothing throws across the boundary; the message for the
 * last failure on the calling thread is gcwt_last_error().  The caller owns all
 * buffers it passes in.  A plan owns its device workspace and stream and must be
 * used from one host thread at a time; different plans are independent.
 */
#ifndef GHOSTCWT_H
#define GHOSTCWT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCWT_ABI_VERSION 5

typedef enum {
  GCWT_OK = 0,
  GCWT_ERR_INVALID = -1,     /* bad argument (the reference raises ValueError)   */
  GCWT_ERR_UNSUPPORTED = -2, /* valid request outside what this build handles     */
  GCWT_ERR_NO_DEVICE = -3,   /* no HIP device / HIP runtime failure at init       */
  GCWT_ERR_HIP = -4,         /* a HIP call failed; text in gcwt_last_error()      */
  GCWT_ERR_NOMEM = -5,
  GCWT_ERR_COMM = -6,        /* RCCL unavailable, or a collective could not be ENQUEUED (bad
                                communicator / library state: the same on every rank)          */
  GCWT_ERR_COMM_INCOMPLETE = -7 /* a collective was enqueued and did not complete: a peer rank
                                is gone (or the device failed under it); the communicator is
                                unusable                                                       */
} gcwt_status;

/* What transform() keeps per coefficient.  The reference keeps abs() only
 * (transforms.py:204); power is np.square of it (transforms.py:510). */
typedef enum {
  GCWT_OUT_AMPLITUDE_F32 = 0, /* |W|       float32 [C][S][N]                     */
  GCWT_OUT_POWER_F32 = 1,     /* |W|^2     float32 [C][S][N]                     */
  GCWT_OUT_COMPLEX_C64 = 2    /* W         float32 pairs (re,im) [C][S][N]       */
} gcwt_out_mode;

/* How a scale is evaluated (reported by gcwt_plan_scale_info).  Every method computes
 * the convolution with the reference's own L-tap kernel (morseutils.py:117-149): the
 * choice is made per scale from the measured support of that kernel's exact response. */
typedef enum {
  GCWT_SCALE_SPECTRAL = 0, /* exact response on a decimated band, block IFFT (fast path) */
  GCWT_SCALE_DIRECT = 1,   /* literal L-tap kernel, time domain: short kernels whose
                              response reaches Nyquist           */
  GCWT_SCALE_FULLBAND = 2, /* one FFT convolution over the whole band: kernels whose
                              truncation leaks at every frequency (small beta)   */
  GCWT_SCALE_BLOCKCONV = 3 /* the same convolution by overlap-save: 4096-sample blocks of
                              the recording, all scales of a group per block while its
                              spectrum sits in registers (kernels up to 2560 taps)    */
} gcwt_scale_method;

enum {
  GCWT_X_ON_DEVICE = 1,   /* gcwt_execute: x is a device pointer                 */
  GCWT_OUT_ON_DEVICE = 2, /* gcwt_execute: out is a device pointer               */
  GCWT_REUSE_MEANS = 4,   /* gcwt_execute_block: keep the channel means of the
                             previous call on this plan (same x)                 */
  GCWT_OUT_F64 = 8,       /* host output only: out is float64 (amplitude, power) or
                             float64 pairs (complex), the reference's result dtype
                             (transforms.py:185); widened while the copy is in flight */
  GCWT_HOST_PINNED = 16   /* gcwt_rows_to_host: dst is page-locked (gcwt_host_alloc): the copy goes
                             straight into it, no staging                         */
};

typedef struct gcwt_plan gcwt_plan;

typedef struct {
  int64_t n_samples;           /* N, samples per channel                          */
  int32_t n_channels;          /* C                                               */
  int32_t n_freqs;             /* S                                               */
  double fs;                   /* Hz                                              */
  double gamma, beta;          /* Morse parameters (morse.py:45-51)               */
  const double* freqs_hz;      /* [S] analysis (peak) frequencies, any order      */
  int32_t n_epochs;            /* E >= 1                                          */
  int32_t out_mode;            /* gcwt_out_mode                                   */
  const int64_t* epoch_bounds; /* [2E] start,stop index pairs (transforms.py:202) */
  int32_t device;              /* HIP ordinal; -1 = leave the current device      */
  int32_t block;               /* decimated block length B; 0 = default (256)     */
  double band_eps;             /* response the fast path may ignore outside a scale's
                                  band, relative to the peak; 0 = 2e-7             */
  int32_t max_fft_log2;        /* longest FFT the plan may use, 12..24; 0 = 22, or 23 / 24 when the
                                  longest kernel does not fit time blocks of 2^22 samples (below
                                  0.13 Hz at 30 kHz).  Epochs that need more are cut into overlapping
                                  time blocks                                                          */
  int32_t wavelet_flags;       /* 0 = what transform() uses: first wavelet, 'bandpass'
                                  (morse.py:84-91).  Bits 0-7: order k of the orthogonal
                                  family (0 = first; morseutils.py:181-196); bit 8: 'energy'
                                  normalisation (morseutils.py:119-124, 186-189)         */
  int32_t precision;           /* gcwt_precision; 0 = default (high)                                     */
  int32_t reserved0;           /* must be 0                                                              */
  double support_tol;          /* share of a kernel's energy (L2, relative) a block halo may cut off;
                                  0 = 4.5e-6                                                              */
} gcwt_params;

#define GCWT_WAVELET_ENERGY 0x100

/* The reference computes in float64 (transforms.py:142-143, convolution.py:68-77).  HIGH (the
 * default): x - mean and the forward FFT of every epoch in float64, and every decimation level cuts
 * its slice of the spectrum to zero below the band of its scales (where every gain is under 2e-8 of
 * its peak) before the float32 stages: content BELOW the analysed bands (drift, offsets, 1/f^n
 * backgrounds) up to ~1000 x the quietest band still meets 1e-5 (FAST: ~65 x).  Interference INSIDE a
 * level's band (a mains line between its scales) is good to ~65 x in both (profiles/r04_dynamic_range.md);
 * EXACT below lifts both limits at 3.4 x the time.  FAST: float32 throughout (rounds 1-3). */
typedef enum {
  GCWT_PRECISION_DEFAULT = 0,
  GCWT_PRECISION_FAST = 1,
  GCWT_PRECISION_HIGH = 2,
  GCWT_PRECISION_AUTO = 4,   /* what DEFAULT means since ABI 5: HIGH, watched -- while the float64 spectrum is made,
                              * every scale's loss to the float32 stages of its decimation level is predicted from
                              * the band energies of that spectrum (the level's content against the scale's own), and
                              * the scales predicted above 1.5e-6 of their peak (a mains line inside an analysed band at
                              * more than ~10 x the recording's spread) are made again by EXACT's paths, the others
                              * keep the fast one: the reference's float64 dynamic range (transforms.py:142-143,
                              * convolution.py:68-77) without asking for it.  gcwt_plan_precision_report tells what
                              * happened.  HIGH itself predicts and reports but never reroutes. */
  GCWT_PRECISION_EXACT = 3   /* no decimated path: every scale through the block convolution (kernels up to 1024
                              * taps, block edges faded) or the full-band path -- float64 forward transforms, and
                              * the float32 stages after them only ever see what the scale's own filter lets
                              * through, so an error stays relative to the scale's own output whatever else the
                              * recording holds: a mains line or a drift 1e4 x the recording's std reads 5e-7 /
                              * 2e-6 (profiles/r04_dynamic_range.md).  3.4 x the time of HIGH on the default
                              * wavelet at the headline shape. */
} gcwt_precision;

typedef struct {
  int32_t abi_version;
  int32_t n_levels;            /* distinct decimation factors in use              */
  int32_t n_spectral, n_direct;
  int32_t block;               /* B                                               */
  int32_t max_decimation;      /* largest R                                       */
  int64_t fft_length;          /* P of the longest epoch                          */
  int64_t workspace_bytes;     /* device workspace the plan will allocate         */
  int64_t out_bytes;           /* size of the out buffer gcwt_execute fills       */
  int32_t n_fullband;          /* scales on the full-band path                    */
  int32_t n_interp;            /* of n_spectral: scales made by the interpolating
                                  synthesis (amplitude / power, R >= 16)          */
  int32_t n_blockconv;         /* scales on the block-convolution path            */
  int32_t reserved;
} gcwt_plan_info;

/* Stage timings of the last gcwt_execute on a plan created with profiling on,
 * in milliseconds, measured with HIP events on the plan's stream. */
typedef struct {
  float mean_ms;        /* per-channel mean (transforms.py:143)                   */
  float fwd_fft_ms;     /* forward FFT of every epoch                             */
  float decimate_ms;    /* per-level inverse FFTs producing x_R                   */
  float block_fft_ms;   /* block spectra                                          */
  float synth_ms;       /* fused filter * twiddle * IFFT * |.| * store  (dominant)*/
  float direct_ms;      /* time-domain scales                                     */
  float total_ms;       /* first kernel start to last kernel end                  */
  int32_t synth_launches;
  float fullband_ms;    /* full-band scales (filter, product, inverse FFT, store) */
  float interp_ms;      /* of synth_ms: the interpolating synthesis kernel        */
  float blockconv_ms;   /* block-convolution scales (block spectra, scales)        */
} gcwt_timings;

/* Library / device ------------------------------------------------------- */
int gcwt_abi_version(void);
const char* gcwt_last_error(void);
int gcwt_device_count(int* count);
int gcwt_device_name(int device, char* buf, size_t buflen);
/* PCI bus id of a device ("0000:c1:00.0"): a launcher looks its NUMA node up in sysfs to pin the
 * rank's host threads beside its GPU. */
int gcwt_device_pci_bus_id(int device, char* buf, size_t buflen);
int gcwt_set_device(int device);
int gcwt_current_device(int* device);
int gcwt_device_malloc(void** ptr, size_t bytes);
int gcwt_device_free(void* ptr);
int gcwt_memcpy_h2d(void* dst, const void* src, size_t bytes);
int gcwt_memcpy_d2h(void* dst, const void* src, size_t bytes);
int gcwt_device_memset(void* dst, int value, size_t bytes);
int gcwt_device_synchronize(void);
/* Free and total bytes of the current device's memory (for sizing time blocks). */
int gcwt_device_memory(size_t* free_bytes, size_t* total_bytes);
/* Page-locked host memory for results: a destination the device's copy engines write at the link's
 * rate, with no staging copy and no first-touch page faults (what a fresh array of the reference's
 * result size, transforms.py:185, costs on the host). */
int gcwt_host_alloc(void** ptr, size_t bytes);
int gcwt_host_free(void* ptr);
/* A rectangle of a device-resident result (gcwt_execute with GCWT_OUT_ON_DEVICE) to the host: n_rows rows of
 * row_elems float32 (complex results: two per sample), src_pitch elements apart on the device, to rows dst_pitch
 * elements apart in dst.  flags: GCWT_OUT_F64 -- dst is float64, the reference's result dtype (transforms.py:185,
 * 203-204): float32 over the link, widened by a standing pool of host threads; GCWT_HOST_PINNED -- dst is
 * page-locked (else it goes through a pinned staging ring and dst_pitch must equal row_elems).  What ContinuousWaveletTransform.amplitude does on first
 * access, and how any (scale, sample) range of a result is read without moving the rest (transforms.py:496-527). */
int gcwt_rows_to_host(const float* d_src, int64_t src_pitch, int64_t n_rows, int64_t row_elems, void* dst,
                      int64_t dst_pitch, int flags);

/* Planning: host only, touches no device.  Replaces the per-call setup of
 * transforms.py:179-185 (wavelet lengths, output allocation) and decides, per
 * scale, the decimation factor and block layout. */
int gcwt_plan_create(gcwt_plan** out, const gcwt_params* params);
void gcwt_plan_destroy(gcwt_plan* plan);
int gcwt_plan_get_info(const gcwt_plan* plan, gcwt_plan_info* info);
/* Per-scale arrays of length S; any pointer may be NULL.  length[] is the
 * reference kernel length L of morse.py:108-122. */
int gcwt_plan_scale_info(const gcwt_plan* plan, int32_t* method, int32_t* decimation,
                         int32_t* halo, int32_t* hop, int64_t* length);
/* What the planner measured on each scale's exact response (arrays of length S, any may
 * be NULL): theta_hi, the radian frequency above which (through the negative
 * frequencies, up to 2 pi) the response stays below band_eps of its peak; support, the
 * distance in samples from the kernel's centre that a block halo must cover; n_bins, the
 * spectrum samples of morseutils.py:130-131 the kernel is built from. */
int gcwt_plan_scale_support(const gcwt_plan* plan, double* theta_hi, double* support,
                            int32_t* n_bins);
/* Stage timings (gcwt_get_timings; the reference's `verbose` print, transforms.py:226-229).  enabled = 1: every stage of
 * an execute lies between two HIP events on its stream; 2: only the synthesis kernels do (synth_ms, interp_ms and
 * synth_launches are filled, total_ms is their span, the other stages read 0) -- the events themselves cost a step of the
 * headline shape 0.19 ms of 13.5, so a loop that is itself being timed asks for level 2; 0: none. */
int gcwt_plan_set_profiling(gcwt_plan* plan, int enabled);
/* After an execute of a plan with precision DEFAULT / AUTO / HIGH: predicted[s] (S floats, may be NULL) is the
 * predicted loss of scale s to the float32 stages of its decimation level, relative to its own output (0 for
 * scales on the exact paths), *worst the largest of them, *n_rerouted how many scales that execute made again by
 * the exact paths (AUTO; 0 for HIGH).  Negative: that many scales were over the threshold and some of them could NOT
 * be made again -- no device memory for the exact sub-plan (gcwt_last_error tells) -- so their rows are the fast
 * path's, what HIGH returns; the execute itself succeeded.  The reference needs no such thing: it is float64 end
 * to end (transforms.py:142-143). */
int gcwt_plan_precision_report(const gcwt_plan* plan, float* predicted, float* worst, int32_t* n_rerouted);
/* Row pitch, in samples, of DEVICE output buffers (0 = dense rows of N, or of the block
 * length).  Rows whose byte offset is not a multiple of 128 make every store straddle
 * cache lines; pad rows to a multiple of 32 samples for full speed when N is not one.
 * Host output buffers are always dense (the library pads internally). */
int gcwt_plan_set_row_pitch(gcwt_plan* plan, int64_t pitch_samples);

/* Device side ------------------------------------------------------------ */
/* Allocate workspace, build the Morse filter bank and FFT tables on the
 * device.  Called implicitly by the first gcwt_execute.  Replaces the kernel
 * construction of morse.py:84-91 / morseutils.py:93-198 for every scale. */
int gcwt_plan_upload(gcwt_plan* plan);

/* The transform: transforms.py:142-143 (global mean), :187-204 (per scale, per
 * epoch convolution + abs) for every channel.  x: float32 [C][N] row-major.
 * out: per out_mode, [C][S][N] row-major, scales in the order of freqs_hz.
 * Samples outside every epoch are written as 0 (transforms.py:185). */
int gcwt_execute(gcwt_plan* plan, const void* x, void* out, int flags);

/* Streaming form of the same transform for recordings whose output does not fit
 * in memory: computes samples [start, start+length) of every channel and scale.
 * x is the WHOLE recording, float32 [C][N] (the global mean and the kernels'
 * support around the range are taken from it); out is [C][S][length].  Cheapest
 * when ranges coincide with the plan's time blocks (gcwt_plan_segment_info).
 * This is the overlap-save front end the reference sketches in
 * ghost/wave/transforms.py:529-597 (_cwtft: chunks with both edges discarded). */
int gcwt_execute_block(gcwt_plan* plan, const void* x, void* out, int64_t start, int64_t length,
                       int flags);
/* Time blocks the plan cuts the epochs into: output range [core_start, core_stop). */
int gcwt_plan_segment_count(const gcwt_plan* plan);
int gcwt_plan_segment_info(const gcwt_plan* plan, int segment, int64_t* core_start,
                           int64_t* core_stop, int64_t* fft_length);

/* Filter bank as held on the device, for parity tests of the bank builder:
 * bank: float32 (re,im) [S][B], H_s(2 pi k / (B R_s)), zero rows for direct
 * scales.  Compare with the spectrum of morseutils.py:130-131. */
int gcwt_filter_bank(gcwt_plan* plan, float* bank);
/* Literal kernel of a direct scale: float32 (re,im) [L].  Compare with psi of
 * morseutils.py:149. */
int gcwt_direct_kernel(gcwt_plan* plan, int scale, float* psi);

int gcwt_get_timings(const gcwt_plan* plan, gcwt_timings* t);

/* The operator layer under the transform: FFT convolution of real signals with one kernel,
 * ghost/sigtools/convolution.py:16-87 (fastconv_scipy) / :89-216 (fastconv_fftw) for a
 * kernel in the time domain, :218-402 (fastconv_freq_scipy / _fftw) for a kernel given by
 * its DFT.  A plan is made once per (signal length n, kernel length m, channel count) and
 * reused: it owns its stream, FFT tables, the kernel's spectrum and the workspace.  Signals
 * longer than one FFT run as overlap-save chunks (the reference's chunked overlap-add,
 * convolution.py:68-77); chunks and channels are batched.  fft_log2: 0 = the smallest
 * 2^k >= n + 2 (m - 1) -- one overlap-save chunk (m - 1 samples of history, then the signal and
 * its tail) holds the whole convolution -- at most 2^22; or 12..22 to fix the FFT length F = 2^fft_log2 (the
 * reference's fft_length: chunks of F - m + 1 samples, convolution.py:70). */
typedef struct gcwt_conv_plan gcwt_conv_plan;
int gcwt_conv_plan_create(gcwt_conv_plan** out, int64_t n, int64_t m, int32_t n_channels,
                          int32_t fft_log2, int32_t device);
void gcwt_conv_plan_destroy(gcwt_conv_plan* plan);
int gcwt_conv_plan_info(const gcwt_conv_plan* plan, int64_t* fft_length, int64_t* chunk,
                        int64_t* n_chunks);
/* kernel: float32 [m] real, or (re,im) pairs [m] when is_complex; host or device memory. */
int gcwt_conv_plan_set_kernel(gcwt_conv_plan* plan, const float* kernel, int is_complex,
                              int on_device);
/* kernel_fd: (re,im) pairs, the DFT of the (zero-padded) kernel on the plan's own
 * fft_length-point grid, natural bin order (the kernel_fd argument of
 * convolution.py:218-219 when its length is a power of two the plan can take). */
int gcwt_conv_plan_set_kernel_fd(gcwt_conv_plan* plan, const float* kernel_fd, int on_device);
/* signal: float32 [C][n]; out: (re,im) pairs [C][count], count = n+m-1 ('full', mode 0),
 * n ('same', mode 1, centred as convolution.py:85) or n-m+1 ('valid', mode 2).
 * flags: GCWT_X_ON_DEVICE, GCWT_OUT_ON_DEVICE. */
int gcwt_conv_plan_execute(gcwt_conv_plan* plan, const float* signal, int mode, float* out,
                           int flags);
/* One-shot form: one signal, one kernel, host memory (a plan made and destroyed inside). */
int gcwt_fastconv(const float* signal, int64_t n, const float* kernel, int64_t m,
                  int kernel_is_complex, int mode, float* out, int device);

/* DFT of arbitrary length n (not only powers of two) by the chirp-z identity, the operator
 * of ghost/sigtools/fourier.py:9-52 (chirpz_dft).  x: float32 [n] real, or (re,im) pairs
 * when is_complex; inverse != 0 gives the normalised inverse DFT.  out: (re,im) pairs [n]
 * (host).  n <= 2^21. */
int gcwt_dft(const float* x, int64_t n, int is_complex, int inverse, float* out, int device);

/* Analytic signal x + i*Hilbert(x) of a real signal through an fft_length-point DFT
 * (0 = n, the reference's default: no padding), as ghost/sigtools/analytic.py:22-112
 * (analytic_signal_fftw; the same numbers as scipy.signal.hilbert(x, N=fft_length)[:n]).
 * signal: float32 [n] (host); out: (re,im) pairs [n] (host).  fft_length <= 2^21, any
 * length (the DFTs run as chirp-z transforms on power-of-two FFTs). */
int gcwt_analytic_signal(const float* signal, int64_t n, int64_t fft_length, float* out,
                         int device);

/* The same operators in float64, the reference's own arithmetic and result dtype (complex128 from float64 FFTs:
 * ghost/sigtools/convolution.py:68-87, fourier.py:9-52, analytic.py:22-112) -- for callers that compare element by
 * element, down to a result's zero crossings.  Host memory in and out; out: (re, im) pairs of doubles.
 * gcwt_dft_f64: any n <= 2^23 (powers of two directly, other lengths by the chirp-z identity, phases reduced in
 * integers).  gcwt_fastconv_f64: signal and kernel real or complex; mode 0 'full', 1 'same' (centred as
 * convolution.py:85), 2 'valid'; results beyond 2^24 samples are made by overlap-add over chunks of the signal like
 * convolution.py:70-77's (kernels up to 2^23 taps then).  The analytic signal is two gcwt_dft_f64 calls around the
 * one-sided mask (ghost_amd.sigtools.analytic_signal_hip(precision='high')). */
int gcwt_dft_f64(const double* x, int64_t n, int is_complex, int inverse, double* out, int device);
int gcwt_fastconv_f64(const double* signal, int64_t n, int signal_is_complex, const double* kernel, int64_t m,
                      int kernel_is_complex, int mode, double* out, int device);

/* Multi-GPU control plane (one process per GPU; RCCL over xGMI).  The data path
 * has no collective: channels are sharded.  The filter bank is broadcast once
 * from rank 0 as BASELINE.json asks; barrier/all-reduce exist for bench timing. */
typedef struct gcwt_comm gcwt_comm;
#define GCWT_COMM_ID_BYTES 128
int gcwt_comm_unique_id(void* id128);
int gcwt_comm_create(gcwt_comm** out, int rank, int n_ranks, const void* id128);
void gcwt_comm_destroy(gcwt_comm* comm);
/* Tears a communicator down after a failed collective without waiting for anything in flight
 * (ncclCommAbort; its stream and scratch are left to process exit).  gcwt_comm_destroy is for
 * communicators whose operations have all completed. */
void gcwt_comm_abort(gcwt_comm* comm);
int gcwt_comm_barrier(gcwt_comm* comm);
int gcwt_comm_allreduce_max(gcwt_comm* comm, double* value);
int gcwt_comm_broadcast_bank(gcwt_comm* comm, gcwt_plan* plan, int root);

#ifdef __cplusplus
}
#endif
#endif /* GHOSTCWT_H */
