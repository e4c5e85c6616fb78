// synthp.hip -- pipelined interpolating synthesis: the work of synthi.hip's k_synthi (q = 2 phases of the
// 256-point inverse transform per (block, scale), demodulated; 8-tap polyphase FIR; |.|; 1 KB stores per wave)
// with the two halves given to DIFFERENT waves of one 512-thread workgroup and a double-buffered z:
//
//   producers (waves 0-1)  gain x P, DFT16, W256 twiddle, a WAVE-LOCAL 16 x 16 exchange, DFT16 -> z of round n + 1
//   consumers (waves 2-7)  FIR + |.| + stores of round n
//
// and ONE workgroup barrier per round (k_synthi: four per pass, and its pass A -- 8 % to 33 % of a pass -- ran
// with nothing of the workgroup storing: profiles/r03_synth_study.md 2c).  A round is 8 columns = 4 z slots =
// nb blocks x ns scales (nb ns = 4); a producer lane keeps its (block, scale slot, phase) for the whole walk,
// so P never changes and the gains of the next round come straight from L2 into its registers: producers never
// store, so their loads never queue behind stores (vmcnt counts both on this ISA).  The half-sample delay of
// even kernel lengths goes into P when a lane's walk reaches those scales (k_synth7's way), so
// every row is interpolated with the tau = rho / I table and the sub-sample position rho = 0 is z itself: with
// I = 4 (R = 8) a lane's first sample needs no FIR (EXACT0).
// The nb blocks of a round are consecutive, so a (round, scale)'s output is one run of nb hop R samples: wave-tasks
// of 256 samples run through it without regard to block edges (a lane picks its block's z slot).
// (transforms.py:203-204: convolve each epoch with each scale's kernel, keep abs.)
#include <hip/hip_runtime.h>

#include "interp.h"
#include "kernels.h"
#include "synth_math.h"

#ifndef GCWT_STORE_AUX
#define GCWT_STORE_AUX 2   // nt: the rows are written once and not read by this launch
#endif

namespace gcwt {

namespace {
constexpr int kT = kInterpTaps;
static_assert(kT == 8, "the FIR loop below is written for 8 taps");
constexpr int kPT = kSynthpThreads;                 // 512
constexpr int kProd = kSynthpProducers;             // producer waves
constexpr int kCons = kPT / 64 - kProd;             // consumer waves
constexpr int kZS = 512 + 16;                       // v2f per z slot: 256 samples x 2 phases, padded so that the two
                                                    // slots a producer wave writes fall on different banks
constexpr int kZBuf = kSynthpSlots * kZS;           // one z buffer
constexpr int kExPS = 66;                           // exchange plane stride (v2f): 16 planes x 2 banks apart
constexpr int kExWave = 16 * kExPS;
constexpr int kLdsP = (2 * kZBuf + kProd * kExWave + 256) * 8 + 2 * 256 * 4 + kSynthpMaxFactor * kInterpTaps * 4;
static_assert(kProd * 4 == 2 * kSynthpSlots, "a producer wave makes four columns = two z slots (q = 2)");

__device__ __forceinline__ v2f fir_mul_lo(v2f z, v2f c) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(z), "v"(c));
  return r;
}
__device__ __forceinline__ v2f fir_fma_lo(v2f acc, v2f z, v2f c) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(z), "v"(c), "v"(acc));
  return r;
}
__device__ __forceinline__ v2f fir_fma_hi(v2f acc, v2f z, v2f c) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(z), "v"(c), "v"(acc));
  return r;
}
}  // namespace

template <int MODE, bool EXACT0>
__global__ void __launch_bounds__(kPT, 4) k_synthp(const SynthpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const zb = reinterpret_cast<v2f*>(smem);                  // [2][kSynthpSlots][kZS]
  v2f* const exb = zb + 2 * kZBuf;                               // [kProd][16][kExPS]
  v2f* const twl = exb + kProd * kExWave;                        // exp(+2 pi i n / 256)
  int* const sc_lds = reinterpret_cast<int*>(twl + 256);
  int* const aux_lds = sc_lds + 256;
  float* const coef_lds = reinterpret_cast<float*>(aux_lds + 256);   // [I][8]: the level's tau = rho / I interpolators

  const SynthpItem it = a.items[a.channels_fastest ? blockIdx.y : blockIdx.x];
  const SynthpLevel lv = a.levels[it.level];
  const int c = a.channels_fastest ? blockIdx.x : blockIdx.y;   // workspace slot: segment * n_channels + channel
  const int seg = c / a.seg.n_channels, ch = c - seg * a.seg.n_channels;
  const int R = lv.decimation, hop = lv.hop, halo = lv.halo, I = lv.factor;
  const int lgnb = lv.log2nb, nb = 1 << lgnb;
  const int lgns = 2 - lgnb, ns = 1 << lgns;                     // nb ns = kSynthpSlots = 4
  const int hopR = hop * R;
  const int64_t n_b = (int64_t)(lv.blk_base + it.blk0) * hopR;   // first kept sample of the first block
  const int64_t w_lo = a.seg.w_lo[seg];
  const int64_t w_len = a.seg.w_hi[seg] - w_lo;
  // the level grids are the union over the batch's segments: nothing of these blocks inside the
  // segment's window -> leave (workgroup-uniform)
  if (n_b + (int64_t)nb * hopR <= w_lo || n_b >= w_lo + w_len) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_scales = lv.n_scales;
  const int* const scales = a.scale_list + lv.scale_offset;
  const int* const auxs = a.scale_aux + lv.scale_offset;
  for (int i = tid; i < n_scales; i += kPT) { sc_lds[i] = scales[i]; aux_lds[i] = auxs[i]; }
  if (tid < 256) {
    const float2 w = a.tw256[tid];
    twl[tid] = (v2f){w.x, w.y};
  }
  for (int i = tid; i < I * kT; i += kPT) coef_lds[i] = a.coef[lv.coef_offset + i];
  // Block spectra XB = FFT_256(x_R[(b hop - halo + n) mod M]) / (256 P), 16 threads per block, as
  // conj(IFFT(conj .)) on the packed inverse DFT16 (k_synth7's prologue); exchange in z buffer 0, spectra
  // in z buffer 1 (the first round's z goes to buffer 0 after every producer has built its P)
  v2f* const fx = zb;
  v2f* const xbs = zb + kZBuf;
  static_assert(4 * 256 <= kZBuf, "the prologue's buffers fit a z buffer");
  {
    const int colw = tid >> 4, t = tid & 15;
    v2f v[16];
    if (colw < nb) {
      // blocks past the level's last one reuse it and are never stored
      const int64_t base = (int64_t)(lv.blk_base + min(it.blk0 + colw, lv.nblk - 1)) * hop - halo + t;
      const float2* xr = a.xr + (int64_t)c * a.xr_cstride + lv.xr_offset;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 u = xr[(base + 16 * j) & lv.m_mask];
        v[j] = (v2f){u.x, -u.y};
      }
      idft16v(v);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float2 w = a.tw256[(t * m) & 255];
        fx[colw * 256 + t * 16 + (m ^ t)] = cmulv(v[dft16_pos(m)], (v2f){w.x, w.y});
      }
    }
    __syncthreads();
    if (colw < nb) {
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = fx[colw * 256 + k1 * 16 + (t ^ k1)];
      idft16v(v);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const v2f z = v[dft16_pos(j)];
        xbs[colw * 256 + t + 16 * j] = (v2f){z.x * a.xb_scale, -z.y * a.xb_scale};
      }
    }
    __syncthreads();
  }
  const int round_end = it.round0 + it.n_rounds;

  // ---- what every wave needs to interpolate, take |.| and store ---------------------------------------------------
  int lgi4 = 0;
  while ((4 << lgi4) < I) ++lgi4;                                // I / 4 = 1 << lgi4 (<= 64: a lane's sub-sample
                                                                 // positions depend on the lane alone)
  const int nb_ok = min(nb, lv.nblk - it.blk0);                  // blocks past the level's last one are not stored
  const int hop4 = hopR >> 2;                                    // lane-tasks (4 samples) per block
  const int tps = nb_ok * hop4;                                  // ... of a (round, scale) run
  const unsigned ext_bytes = w_len > 0 ? (unsigned)(w_len * 4) : 0u;
  float* const out0 = a.out + ((int64_t)ch * a.n_scales * a.row_len + a.seg.seg_col[seg] + w_lo);
  const int s_base = (int)(n_b - w_lo);                          // window-relative sample of the run's first
  // wave-tasks (256 samples) with a sample inside this launch's window
  const int wt_a = s_base >= 0 ? 0 : (-s_base) >> 8;
  const int wt_b = (int)min((int64_t)((tps + 63) >> 6), max((int64_t)0, (w_len - s_base + 255) >> 8));
  const int n_t = max(0, wt_b - wt_a);
  const int zc0 = (halo << 1) - (kT / 2 - 1);                    // z index of the window of lane-task 0, block 0
  const int zc_blk = ns * kZS - 2 * hop;                         // what one block further adds to it
  // A round's tasks -- (scale slot, wave-task of 256 samples), numbered slot-major -- are shared out by a fixed rule:
  // the consumer waves take tasks [0, n_cons) in turn, the producer waves, once the next round's z is made, the
  // remaining total * help / 128 (a figure per level from the host: how much of a round a producer has left).
  // The z window of a wave's next task is read from LDS while the current one is worked (two register sets).
  const int help = lv.help;
  const int s_hop4 = __builtin_amdgcn_readfirstlane(hop4);
  auto consume = [&](int round, const v2f (&cf)[4][kT / 2], int first, int stride, bool helper) {
    const int par = (round - it.round0) & 1;
    const v2f* const zcur = zb + par * kZBuf;
    const int n_sl = min(ns, n_scales - round * ns);
    const int total = n_sl * n_t;
    const int n_help = (total * help) >> 7;
    const int t_lo = helper ? total - n_help : 0, t_hi = helper ? total : total - n_help;
    // the z window of task (slot sl, wave-task wt) for this lane: 8 consecutive z of its block's slot
    auto load_z = [&](v2f (&z)[kT], int sl, int wt) {
      const int k0 = __builtin_amdgcn_readfirstlane(wt * 64);
      int zoff = (k0 + lane) >> lgi4;
      int zbase = sl * kZS + zc0;                                // scalar
      if (nb > 1) {                                              // workgroup-uniform
        const int bl0 = __builtin_amdgcn_readfirstlane((k0 >= s_hop4 ? 1 : 0) + (k0 >= 2 * s_hop4 ? 1 : 0) + (k0 >= 3 * s_hop4 ? 1 : 0));
        const int bnd = (bl0 + 1) * s_hop4;                      // the wave-task may run into the next block
        zbase += bl0 * zc_blk;
        zoff += k0 + lane >= bnd ? zc_blk : 0;
      }
      const v2f* const z2 = zcur + zbase + zoff;
#pragma unroll
      for (int j = 0; j < kT; ++j) z[j] = z2[j];
    };
    int sl_cur = -1;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out0, 0, 0, 0x00020000);
    // first half: the slot's store descriptor when the slot changes, taps 0 .. 3; second half: the rest, |.|, store.
    // The next task's z window is asked for between the two, so that it has half a task's time to arrive and no
    // younger LDS read is outstanding when it is waited for (LDS results return in order).
    v2f acc[4];
    auto work_a = [&](const v2f (&z)[kT], int sl) {
      if (sl != sl_cur) {                                        // wave-uniform
        sl_cur = sl;
        const int entry = __builtin_amdgcn_readfirstlane(sc_lds[round * ns + sl]);
        // descriptor from provably wave-uniform words (else hipcc waterfalls every store); it spans
        // exactly the samples this launch may write, [w_lo, w_hi) of the segment
        const int srow = entry & kScaleIndexMask;
        const uint64_t dst_bits = reinterpret_cast<uint64_t>(out0 + (int64_t)srow * a.row_len);
        const uint32_t dst_lo = __builtin_amdgcn_readfirstlane((uint32_t)dst_bits);
        const uint32_t dst_hi = __builtin_amdgcn_readfirstlane((uint32_t)(dst_bits >> 32));
        float* const dst = reinterpret_cast<float*>(((uint64_t)dst_hi << 32) | dst_lo);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, __builtin_amdgcn_readfirstlane(ext_bytes), 0x00020000);
      }
#pragma unroll
      for (int i = EXACT0 ? 1 : 0; i < 4; ++i) acc[i] = fir_fma_hi(fir_mul_lo(z[0], cf[i][0]), z[1], cf[i][0]);
#pragma unroll
      for (int i = EXACT0 ? 1 : 0; i < 4; ++i) acc[i] = fir_fma_hi(fir_fma_lo(acc[i], z[2], cf[i][1]), z[3], cf[i][1]);
      if (EXACT0) acc[0] = z[kT / 2 - 1];                        // rho = 0: the interpolator is the unit tap at T/2 - 1
    };
    auto work_b = [&](const v2f (&z)[kT], int wt) {
#pragma unroll
      for (int j = 2; j < kT / 2; ++j)
#pragma unroll
        for (int i = EXACT0 ? 1 : 0; i < 4; ++i) acc[i] = fir_fma_hi(fir_fma_lo(acc[i], z[2 * j], cf[i][j]), z[2 * j + 1], cf[i][j]);
      float res[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p2v = __builtin_fmaf(acc[i].y, acc[i].y, acc[i].x * acc[i].x);
        res[i] = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2v) : p2v;
      }
      const int k0 = __builtin_amdgcn_readfirstlane(wt * 64);
      const int s_first = s_base + 4 * k0;                      // window-relative sample of the wave-task's first
      const bool whole = k0 + 64 <= tps && s_first >= 0 && (int64_t)s_first + 256 <= w_len;
      const int s0 = s_first + 4 * lane;                        // window-relative sample of res[0]
      if (whole) {                                              // wave-uniform: no lane looks at its own range
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u pk = {__builtin_bit_cast(unsigned, res[0]), __builtin_bit_cast(unsigned, res[1]),
                        __builtin_bit_cast(unsigned, res[2]), __builtin_bit_cast(unsigned, res[3])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, (unsigned)s0 * 4u, 0, GCWT_STORE_AUX);
      } else if (k0 + lane < tps) {
        // the window's edge (or the run's last samples) runs through this wave-task: one sample at a time,
        // each under its own test -- four plain stores in a row are merged into one 16-byte store by the
        // compiler, whose single range check then drops the samples inside the window with the ones outside
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((unsigned)(s0 + i) < (unsigned)w_len)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[i]), rsrc,
                                                  (unsigned)(s0 + i) * 4u, 0, GCWT_STORE_AUX);
      }
    };
    // task number -> (slot, wave-task); a wave's tasks come in increasing order, so the slot only moves forward
    int sl = 0, base = 0;
    auto locate = [&](int task) {
      while (task >= base + n_t) { base += n_t; ++sl; }          // scalar; at most ns - 1 steps per round
      return wt_a + (task - base);
    };
    int task = t_lo + first;
    if (task >= t_hi) return;
    v2f za[kT], zn[kT];
    int wt = locate(task), sl_w = sl;
    load_z(za, sl_w, wt);
    for (;;) {
      // (the window of the task after this one is always asked for -- this one's again when there is none -- so that
      // every path into the second half has the same LDS reads in flight: the compiler's waits are then exact)
      const int task2 = task + stride;
      const bool more = task2 < t_hi;
      const int wt2 = more ? locate(task2) : wt, sl2 = sl;
      work_a(za, sl_w);
      load_z(zn, sl2, wt2);
      work_b(za, wt);
      if (!more) break;
      task = task2 + stride;
      const bool more2 = task < t_hi;
      wt = more2 ? locate(task) : wt2;
      sl_w = sl;
      work_a(zn, sl2);
      load_z(za, sl_w, wt);
      work_b(zn, wt2);
      if (!more2) break;
    }
  };
  // a lane's interpolators: its 4 sub-sample positions x 8 taps (the tau = rho / I table)
  auto load_coef = [&](v2f (&cf)[4][kT / 2], const float* src) {
    const float4* const cp = reinterpret_cast<const float4*>(src) + (lane & ((1 << lgi4) - 1)) * (4 * kT / 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 u0 = cp[2 * i], u1 = cp[2 * i + 1];
      cf[i][0] = (v2f){u0.x, u0.y}; cf[i][1] = (v2f){u0.z, u0.w};
      cf[i][2] = (v2f){u1.x, u1.y}; cf[i][3] = (v2f){u1.z, u1.w};
    }
  };

  // The two roles run separate loops (a consumer's registers hold its coefficients, a producer's its P and the next
  // gains); the workgroup barrier counts waves, not call sites, and both loops pass it exactly 1 + n_rounds times.
  if (wave < kProd) {
    // ---- producer: four columns (block, scale slot, phase) of every round, then it helps with the stores ------
    const int colw = lane >> 4, t = lane & 15;
    const int c8 = wave * 4 + colw;
    const int zslot = c8 >> 1, p = c8 & 1;                       // z slot = block * ns + scale slot
    const int sl = zslot & (ns - 1), bl = zslot >> lgns;
    // what never changes for a lane: P[k] = XB[k] W^{k r}, k = t + 16 j, r = p I the phase of its column
    v2f pw[16];
    {
      const int r = p * I;
      const float2* ltw = a.level_tw + lv.tw_offset;
      const float2 b0 = ltw[t * r], st = ltw[16 * r];
      v2f wcur = (v2f){b0.x, b0.y};
      const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        pw[j] = cmulv(xbs[bl * 256 + t + 16 * j], wcur);
        wcur = cmulv(wcur, wstep);
      }
    }
    // second half of the transform: lane (a2, col2) takes output samples a2 + 16 m1 of column col2 of this wave
    const int a2 = lane >> 2, col2 = lane & 3;
    const int c8b = wave * 4 + col2;
    const int zw_off = (c8b >> 1) * kZS + (a2 << 1) + (c8b & 1);        // + 32 m1
    v2f* const exw0 = exb + wave * kExWave + colw;                       // + plane * kExPS + 4 j
    const v2f* const exr = exb + wave * kExWave + lane;                  // + k1 * kExPS
    const float4* const gain_rows = reinterpret_cast<const float4*>(a.gain_lv + (int64_t)lv.scale_offset * 256) + t * 4;
    bool halved = false;
    float4 gq[4];
    int entry = 0, kc = 0, bcur = 0;
    auto fetch = [&](int round) {           // the round's scale of this lane: list entry, demodulation bin, 16 gains
      bcur = min(round * ns + sl, n_scales - 1);
      entry = scales[bcur];
      kc = auxs[bcur] & 0xffff;
#pragma unroll
      for (int i = 0; i < 4; ++i) gq[i] = gain_rows[bcur * 64 + i];
    };
    auto produce = [&](int round, v2f* zdst) {
      if (bcur >= lv.n_plain && !halved) {  // the walk reaches the kernels of even length (lane-wise: the lanes of
        halved = true;                      // a wave may follow different scale slots); once per lane
        const float2* hf = a.level_half_tw + lv.half_offset + t;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float2 h = hf[16 * j];
          pw[j] = cmulv(pw[j], (v2f){h.x, h.y});
        }
      }
      v2f v[16];
      // (16 - j_hi) in the entry's top byte: first-pass inputs j >= j_hi are left out (kernels.h); the
      // smallest cut of the wave's columns, so that the choice is wave-uniform
      unsigned cut = (unsigned)entry >> 24;
      cut = min(min((unsigned)__builtin_amdgcn_readlane((int)cut, 0), (unsigned)__builtin_amdgcn_readlane((int)cut, 16)),
                min((unsigned)__builtin_amdgcn_readlane((int)cut, 32), (unsigned)__builtin_amdgcn_readlane((int)cut, 48)));
      switch (cut) {
#define GCWT_WINDOW(hi) case 16 - (hi): gain_first_layer<hi>(v, pw, gq); break;
        GCWT_WINDOW(15) GCWT_WINDOW(14) GCWT_WINDOW(13) GCWT_WINDOW(12) GCWT_WINDOW(11) GCWT_WINDOW(10) GCWT_WINDOW(9)
#undef GCWT_WINDOW
        default: gain_first_layer<16>(v, pw, gq); break;
      }
      const int kc_now = kc;
      idft16v_tail(v);
      // twiddle W256^{(t - k_c) a - (k_c / 2) p}: bins counted from the demodulation centre; the column's
      // values go to exchange plane (t - k_c) mod 16, so that the second half reads its planes in order
      const unsigned step8 = (unsigned)((t - kc_now) & 255) << 3;
      unsigned idx8 = (unsigned)((-(kc_now >> 1) * p) & 255) << 3;
      v2f* const exw = exw0 + ((t - kc_now) & 15) * kExPS;
      const char* const twb = reinterpret_cast<const char*>(twl);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        exw[4 * j] = cmulv(v[dft16_pos(j)], *reinterpret_cast<const v2f*>(twb + idx8));
        idx8 = (idx8 + step8) & 0x7f8u;
      }
      // the exchange stays inside the wave: its LDS operations retire in order, no barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = exr[k1 * kExPS];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the next round's writes come after these reads)
      __builtin_amdgcn_wave_barrier();
      idft16v(v);
      v2f* const zw = zdst + zw_off;
#pragma unroll
      for (int m1 = 0; m1 < 16; ++m1) zw[32 * m1] = v[dft16_pos(m1)];
      // the next round's scale: on its way while this wave helps with the stores
      if (round + 1 < round_end) fetch(round + 1);
    };
    fetch(it.round0);
    // iteration round0 - 1 makes the first round's z (buffer 0: the prologue's exchange, read out before its last
    // barrier; the spectra sit in buffer 1, which the second round overwrites a barrier after every P is built)
    for (int round = it.round0 - 1; round < round_end; ++round) {
      if (round + 1 < round_end) produce(round + 1, zb + ((round + 1 - it.round0) & 1) * kZBuf);
      if (round >= it.round0 && help > 0) {
        v2f cf[4][kT / 2];                                       // (not kept across produce(): from the table in LDS)
        load_coef(cf, coef_lds);
        consume(round, cf, wave, kProd, true);
      }
      __syncthreads();
    }
  } else {
    // ---- consumer ------------------------------------------------------------------------------------------------
    v2f cf[4][kT / 2];
    load_coef(cf, a.coef + lv.coef_offset);
    __syncthreads();                        // z of the first round is there
    for (int round = it.round0; round < round_end; ++round) {
      consume(round, cf, wave - kProd, kCons, false);
      __syncthreads();                      // z of this round is read out; the next one is complete
    }
  }
}

hipError_t launch_synthp(int mode, const SynthpArgs& a, int n_items, int n_channels, bool exact0, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  if (mode != GCWT_OUT_AMPLITUDE_F32 && mode != GCWT_OUT_POWER_F32) return hipErrorInvalidValue;
  static bool attr_done[64] = {};            // per device: one process may drive several
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_done[dev_ & 63];
  if (!attr_set) {
    const void* fns[4] = {(const void*)k_synthp<GCWT_OUT_AMPLITUDE_F32, false>, (const void*)k_synthp<GCWT_OUT_AMPLITUDE_F32, true>,
                          (const void*)k_synthp<GCWT_OUT_POWER_F32, false>, (const void*)k_synthp<GCWT_OUT_POWER_F32, true>};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsP);
      if (e != hipSuccess) return e;
    }
    attr_set = true;
  }
  if ((a.channels_fastest ? n_items : n_channels) > 65535) return hipErrorInvalidValue;
  const dim3 grid = a.channels_fastest ? dim3(n_channels, n_items) : dim3(n_items, n_channels), block(kPT);
  if (mode == GCWT_OUT_AMPLITUDE_F32) {
    if (exact0) hipLaunchKernelGGL((k_synthp<GCWT_OUT_AMPLITUDE_F32, true>), grid, block, kLdsP, st, a);
    else hipLaunchKernelGGL((k_synthp<GCWT_OUT_AMPLITUDE_F32, false>), grid, block, kLdsP, st, a);
  } else {
    if (exact0) hipLaunchKernelGGL((k_synthp<GCWT_OUT_POWER_F32, true>), grid, block, kLdsP, st, a);
    else hipLaunchKernelGGL((k_synthp<GCWT_OUT_POWER_F32, false>), grid, block, kLdsP, st, a);
  }
  return hipGetLastError();
}

}  // namespace gcwt
