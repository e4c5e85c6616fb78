// synthi.hip -- interpolating synthesis: the amplitude / power rows of the levels whose scales
// are heavily oversampled at the full rate (R >= 16).
// (transforms.py:203-204: convolve each epoch with each scale's kernel, keep abs.)
//
// k_synth7 pays one point of a 256-point inverse FFT for every stored sample.  Here a (block,
// scale) goes through that transform only for q of its R phases (q = 2; 4 / 8 where R / 2 would
// exceed the largest interpolation factor) -- the scale's complex output z at q x the level's
// rate, demodulated to its band centre so that it is a low-pass signal -- and the R / q = I
// samples between two of those come from an 8-tap (q = 2) or 6-tap (q = 4) polyphase FIR with real coefficients
// (interp.h: minimax design under the envelope of the level's gains; what it adds is bounded per
// level from the scales' own gains, planner.cpp: plan_interp_level, and stays below 2e-7 of a
// scale's peak).  |.| does not see the demodulation.  Per stored sample: 8 packed FMAs + |.|
// against ~15 VALU and 3.4 LDS instructions, and the stores are whole 1 KB runs per wave.
//
// One workgroup (256 threads = 16 columns) = one or two blocks of one level; it walks the level's
// scales 16 / (q nb) at a time ("slots").  Per pass:
//   A  16 columns (block, slot, phase): P * G_s, DFT16, W256 twiddle, exchange through LDS, DFT16
//      -- the loop body of k_synth7 -- with the demodulation folded in: bins are counted from the
//      scale's centre k_c (twiddle exponent (t - k_c) a - (k_c / q) p, exchange planes written
//      rotated by k_c), so z needs no multiply of its own.  z lands in LDS in time order.
//   B  every wave takes runs of 256 consecutive output samples of one (block, scale): a lane makes
//      4 consecutive samples from the 8 z values around them (coefficients of its 4 sub-sample
//      positions in registers), |.|, one 16-byte store; a wave store is 1 KB contiguous.
// Kernels of even length carry a half-sample delay: their rows are interpolated at
// tau - 1/(2 I) with a second coefficient table instead of a phase on the spectrum.
// How the work is cut (api.cpp): an item is a block group, a run of passes and a range of the
// wave-tasks of pass B (all of them unless one pass is more than ~2 MB of rows: then several
// workgroups repeat pass A and share B); items are listed largest first.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "interp.h"
#include "kernels.h"
#include "options.h"
#include "synth_math.h"

#ifndef GCWT_STORE_AUX
#define GCWT_STORE_AUX 2   // nt: the rows are written once and not read by this launch
#endif

namespace gcwt {

namespace {
constexpr int kT = kInterpTaps;
static_assert(kT == 8, "the FIR loop below is written for 8 taps");
// Columns per pass.  16 (256 threads, three workgroups and 12 waves per CU, up to 168 VGPRs: the
// persistent operand, the FIR coefficients and both transforms' working sets fit without spills);
// 32 (512 threads, two workgroups and 16 waves per CU, 128 VGPRs) spilled 32 dwords and was slower.
constexpr int kColsI = kInterpCols;
constexpr int kLgColsI = kColsI == 32 ? 5 : kColsI == 16 ? 4 : 3;
constexpr int kThreadsI = 16 * kColsI;
constexpr int kWavesI = kThreadsI / 64;
constexpr int kPlaneI = kThreadsI + 1;
constexpr int kZPad = 4;                         // v2f entries between the slots of the z buffer: their
                                                 // writes fall on different banks
constexpr int kSlotsMax = kColsI / 2;            // z slots and scale slots of a pass: q >= 2
static_assert(kColsI == kInterpCols, "the host cuts the passes for this many columns (interp.h)");
constexpr int kZElems = 256 * kColsI + kSlotsMax * kZPad;     // z buffer
constexpr int kExElems = kZElems > 16 * kPlaneI ? kZElems : 16 * kPlaneI;   // ... in place of the 16 exchange planes
constexpr int kGainRowI = 16 * 20;
constexpr int kLdsBytes = kExElems * 8 + 256 * 8 + kSlotsMax * kGainRowI * 4 + 2 * 256 * 4;

// Two neighbouring samples share an accumulator pair: re = (re_0, re_1), im = (im_0, im_1), the tap's coefficients
// of both samples side by side in one register pair, z's real or imaginary part broadcast to both halves -- so that
// |.|^2 of two samples is one packed multiply and one packed multiply-add.
__device__ __forceinline__ v2f fir_mul_re(v2f z, v2f c) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(z), "v"(c));
  return r;
}
__device__ __forceinline__ v2f fir_mul_im(v2f z, v2f c) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(z), "v"(c));
  return r;
}
__device__ __forceinline__ v2f fir_fma_re(v2f acc, v2f z, v2f c) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(z), "v"(c), "v"(acc));
  return r;
}
__device__ __forceinline__ v2f fir_fma_im(v2f acc, v2f z, v2f c) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(z), "v"(c), "v"(acc));
  return r;
}
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// taps J0 .. J1 - 1 of the table's rows on the z values around a lane's four samples
template <int J0, int J1>
__device__ __forceinline__ void fir_taps(const v2f* zp, const v2f (&cf)[2][kT], v2f (&are)[2], v2f (&aim)[2]) {
  {
    const v2f w = zp[J0];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) { are[pr] = fir_mul_re(w, cf[pr][J0]); aim[pr] = fir_mul_im(w, cf[pr][J0]); }
  }
#pragma unroll
  for (int j = J0 + 1; j < J1; ++j) {
    const v2f w = zp[j];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) { are[pr] = fir_fma_re(are[pr], w, cf[pr][j]); aim[pr] = fir_fma_im(aim[pr], w, cf[pr][j]); }
  }
}
}  // namespace

template <int MODE>
__global__ void __launch_bounds__(kThreadsI, kColsI == 32 ? 4 : 3) k_synthi(const SynthiArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);
  v2f* const twl = ex + kExElems;                                   // exp(+2 pi i n / 256)
  float* const stage = reinterpret_cast<float*>(twl + 256);         // gains of the pass's scales
  int* const sc_lds = reinterpret_cast<int*>(stage + kSlotsMax * kGainRowI);
  int* const aux_lds = sc_lds + 256;

  const SynthiItem it = a.items[a.channels_fastest ? blockIdx.y : blockIdx.x];
  const SynthiLevel lv = a.levels[it.level];
  const int c = a.channels_fastest ? blockIdx.x : blockIdx.y;   // workspace slot: segment * n_channels + channel
  const int seg = c / a.seg.n_channels, ch = c - seg * a.seg.n_channels;
  const int R = lv.decimation, q = lv.q, lgq = lv.log2q, hop = lv.hop, halo = lv.halo;
  const int lgnb = lv.log2nb, nb = 1 << lgnb;                       // blocks per workgroup
  const int64_t n_b = (int64_t)(lv.blk_base + it.blk0) * hop * R;   // first kept sample of the first block
  const int64_t w_lo = a.seg.w_lo[seg];
  const int64_t w_len = a.seg.w_hi[seg] - w_lo;
  // the level grids are the union over the batch's segments: nothing of these blocks inside the
  // segment's window -> leave (workgroup-uniform)
  if (n_b + (int64_t)nb * hop * R <= w_lo || n_b >= w_lo + w_len) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int colw = tid >> 4, t = tid & 15;
  // column -> (block of the group, scale slot of the pass, phase); a "z slot" is a (block, scale) pair
  const int lgns = kLgColsI - lgq - lgnb, ns = 1 << lgns;
  const int zslot = colw >> lgq, p = colw & (q - 1);
  const int slot = zslot & (ns - 1), blk_l = zslot >> lgns;
  const int nzs = kColsI >> lgq;
  const int n_scales = lv.n_scales;
  const int* const scales = a.scale_list + lv.scale_offset;
  const int* const auxs = a.scale_aux + lv.scale_offset;
  for (int i = tid; i < n_scales; i += kThreadsI) { sc_lds[i] = scales[i]; aux_lds[i] = auxs[i]; }
  if (tid < 256) {
    const float2 w = a.tw256[tid];
    twl[tid] = (v2f){w.x, w.y};
  }
  // gains of one pass in LDS, lane t's sixteen values (bins t + 16 j) side by side at a pitch of
  // 20 floats (four conflict-free 16-byte reads per thread), as k_synth7 parks them
  auto stage_slot = [&](int i) { return (i >> 8) * kGainRowI + (i & 15) * 20 + ((i >> 4) & 15); };
  for (int i = tid; i < ns * 256; i += kThreadsI) {
    const int sb = min(it.pass0 * ns + (i >> 8), n_scales - 1);
    stage[stage_slot(i)] = a.gain[(int64_t)(scales[sb] & kScaleIndexMask) * 256 + (i & 255)];
  }

  // Block spectrum XB = FFT_256(x_R[(b hop - halo + n) mod M]) / (256 P), made by 16 threads as
  // conj(IFFT(conj .)) on the packed inverse DFT16 (k_synth7's prologue), left in LDS for all.
  v2f* const fx = ex;
  v2f* const xbs = ex + kSlotsMax * 256;
  {
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
  // what never changes for a thread: P[k] = XB[k] W^{k r}, k = t + 16 j, r = p I the phase of its column
  v2f pw[16];
  {
    const int r = p * lv.factor;
    const float2* ltw = a.level_tw + lv.tw_offset;
    const float2 b0 = ltw[t * r], st = ltw[16 * r];
    v2f wcur = (v2f){b0.x, b0.y};
    const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      pw[j] = cmulv(xbs[blk_l * 256 + t + 16 * j], wcur);
      wcur = cmulv(wcur, wstep);
    }
  }

  // second half of the transform: thread (a2, col2) takes output samples a2 + 16 c of its column
  const int a2 = tid >> kLgColsI, col2 = tid & (kColsI - 1);
  const int zslot2 = col2 >> lgq, p2 = col2 & (q - 1);
  const int zstride = (256 << lgq) + kZPad;
  v2f* const zw = ex + zslot2 * zstride + (a2 << lgq) + p2;   // + 16 q c

  // phase B geometry: lane-tasks of 4 consecutive samples, 64 of them per wave-task
  const int I = lv.factor;
  const bool six = lv.taps == 6;
  int lgi4 = 0;
  while ((4 << lgi4) < I) ++lgi4;                             // I / 4 = 1 << lgi4
  const int tps = hop * (R >> 2);                             // lane-tasks per (block, scale)
  const int n_wt = it.wt_hi - it.wt_lo;                       // wave-tasks per (block, scale) of this workgroup
  // which 4 of the I sub-sample positions a lane-task k covers: k mod (I / 4).  k = 64 wt + lane
  // and a wave's wave-tasks advance by kWavesI = 4 at a time, so up to I / 4 = 256 the set depends
  // on the lane and on wt mod 4 only -- constant along a wave's run through a z slot
  static_assert(GCWT_SYNTHI_COLS != 16 || (kWavesI == 4 && kInterpMaxFactor <= 1024),
                "a wave's lane-tasks keep their class mod I / 4 (the 8- and 32-column measurement builds: I <= 256 only)");
  const float* const coef_lv = a.coef + lv.coef_offset;
  constexpr int kElem = 1;
  const unsigned ext_bytes = w_len > 0 ? (unsigned)(w_len * (4 * kElem)) : 0u;
  float* const out0 = a.out + ((int64_t)ch * a.n_scales * a.row_len + a.seg.seg_col[seg] + w_lo) * kElem;
  const int s_base = (int)(n_b - w_lo);                       // window-relative sample of the block's first
  const int pass_end = it.pass0 + it.n_pass;   // of the (n_scales + ns - 1) >> lgns passes of the level's walk
  int cur_par = -1;                       // which coefficients cf holds: kernels of odd (0) / even (1) length, (wt mod 4 class) << 1
  v2f cf[2][kT];                          // [pair of samples][tap]: the tap's coefficients of samples 2 pair, 2 pair + 1
  __syncthreads();

  for (int pass = it.pass0; pass < pass_end; ++pass) {
    const int b0 = pass * ns;
    const bool has_next = pass + 1 < pass_end;
    // next pass's gains: loaded now (nothing of this pass is in flight yet), parked after the exchange
    float nxt[kSlotsMax];
    if (has_next) {
#pragma unroll
      for (int u = 0; u < kSlotsMax; ++u) {
        const int i = tid + kThreadsI * u;
        if (i < ns * 256) {
          const int sb = min(b0 + ns + (i >> 8), n_scales - 1);
          nxt[u] = a.gain[(int64_t)(sc_lds[sb] & kScaleIndexMask) * 256 + (i & 255)];
        }
      }
    }
    // ---- A: 32 columns through the 256-point inverse transform --------------------------------
    {
      const int bs = min(b0 + slot, n_scales - 1);
      const int entry = sc_lds[bs];
      const int kc = aux_lds[bs] & 0xffff;
      const float4* const hs = reinterpret_cast<const float4*>(stage + slot * kGainRowI + t * 20);
      v2f v[16];
      // (16 - j_hi) in the entry's top byte: first-pass inputs j >= j_hi are left out (kernels.h).
      // A wave's four columns belong to one scale slot (q >= 4) or two (q = 2): the smaller window
      // cut of the two, so that the choice is wave-uniform
      unsigned cut = (unsigned)__builtin_amdgcn_readfirstlane(entry) >> 24;
      if (q == 2) {
        const int other = sc_lds[min(b0 + (((colw ^ 2) >> lgq) & (ns - 1)), n_scales - 1)];
        cut = min(cut, (unsigned)__builtin_amdgcn_readfirstlane(min((unsigned)entry >> 24, (unsigned)other >> 24)));
        cut = (unsigned)__builtin_amdgcn_readfirstlane(cut);
      }
      switch (cut) {
#define GCWT_WINDOW(hi) case 16 - (hi): gain_first_layer<hi>(v, pw, hs); break;
        GCWT_WINDOW(15) GCWT_WINDOW(14) GCWT_WINDOW(13) GCWT_WINDOW(12) GCWT_WINDOW(11) GCWT_WINDOW(10) GCWT_WINDOW(9)
#undef GCWT_WINDOW
        default: gain_first_layer<16>(v, pw, hs); break;
      }
      idft16v_tail(v);
      // twiddle W256^{(t - k_c) a - (k_c / q) p}: bins counted from the demodulation centre; the
      // column's values go to exchange plane (t - k_c) mod 16, so that the second half reads its
      // sixteen planes in order (index arithmetic in bytes: one add and one mask per twiddle)
      const unsigned step8 = (unsigned)((t - kc) & 255) << 3;
      unsigned idx8 = (unsigned)((-(kc >> lgq) * p) & 255) << 3;
      v2f* const exw = ex + ((t - kc) & 15) * kPlaneI + colw;
      const char* const twb = reinterpret_cast<const char*>(twl);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        exw[j * kColsI] = cmulv(v[dft16_pos(j)], *reinterpret_cast<const v2f*>(twb + idx8));
        idx8 = (idx8 + step8) & 0x7f8u;
      }
    }
    __syncthreads();
    if (has_next) {
#pragma unroll
      for (int u = 0; u < kSlotsMax; ++u) {
        const int i = tid + kThreadsI * u;
        if (i < ns * 256) stage[stage_slot(i)] = nxt[u];
      }
    }
    {
      v2f v[16];
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = ex[k1 * kPlaneI + tid];
      __syncthreads();                      // every plane is read before z takes their place
      idft16v(v);
#pragma unroll
      for (int m1 = 0; m1 < 16; ++m1) zw[(16 * m1) << lgq] = v[dft16_pos(m1)];
    }
    __syncthreads();

    // ---- B: interpolate, |.|, store -----------------------------------------------------------
    {
      for (int zi = 0; zi < nzs; ++zi) {                      // everything here is wave-uniform
        const int sl = zi & (ns - 1), bl = zi >> lgns;
        if (b0 + sl >= n_scales || it.blk0 + bl >= lv.nblk) continue;
        const int entry = __builtin_amdgcn_readfirstlane(sc_lds[b0 + sl]);
        const int par = (__builtin_amdgcn_readfirstlane(aux_lds[b0 + sl]) >> 16) & 1;
        const int wt0 = (wave - zi * n_wt) & (kWavesI - 1);
        const int key = par | (lgi4 > 6 ? (wt0 & ((1 << (lgi4 - 6)) - 1)) << 1 : 0);
        if (key != cur_par) {       // rare: the level's list is ordered by parity; wt0 changes only when I > 256
          const int sigma = (wt0 * 64 + lane) & ((1 << lgi4) - 1);
          const float4* const cp = reinterpret_cast<const float4*>(coef_lv + ((int64_t)par * I + sigma * 4) * kT);
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const float4 a0 = cp[4 * pr], a1 = cp[4 * pr + 1], b0 = cp[4 * pr + 2], b1 = cp[4 * pr + 3];
            cf[pr][0] = (v2f){a0.x, b0.x}; cf[pr][1] = (v2f){a0.y, b0.y}; cf[pr][2] = (v2f){a0.z, b0.z}; cf[pr][3] = (v2f){a0.w, b0.w};
            cf[pr][4] = (v2f){a1.x, b1.x}; cf[pr][5] = (v2f){a1.y, b1.y}; cf[pr][6] = (v2f){a1.z, b1.z}; cf[pr][7] = (v2f){a1.w, b1.w};
          }
          cur_par = key;
        }
        // descriptor from provably wave-uniform words (else hipcc waterfalls every store); it spans
        // exactly the samples this launch may write, [w_lo, w_hi) of the segment
        const int srow = entry & kScaleIndexMask;
        const uint64_t dst_bits = reinterpret_cast<uint64_t>(out0 + (int64_t)srow * a.row_len * kElem);
        const uint32_t dst_lo = __builtin_amdgcn_readfirstlane((uint32_t)dst_bits);
        const uint32_t dst_hi = __builtin_amdgcn_readfirstlane((uint32_t)(dst_bits >> 32));
        float* const dst = reinterpret_cast<float*>(((uint64_t)dst_hi << 32) | dst_lo);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            dst, 0, __builtin_amdgcn_readfirstlane(ext_bytes), 0x00020000);
        const v2f* const zs = ex + zi * zstride + (halo << lgq) - (kT / 2 - 1);
        const int s_blk = s_base + bl * hop * R;              // window-relative sample of this block's first
        // the slot's wave-tasks are dealt round-robin over the waves, continuing where the previous
        // slot stopped
        // (only those with a sample inside this launch's window: task wt covers s_blk + 256 wt .. + 255)
        const int wt_a = max(it.wt_lo, s_blk >= 0 ? 0 : (-s_blk) >> 8);
        const int wt_b = (int)min((int64_t)it.wt_hi, max((int64_t)0, (w_len - s_blk + 255) >> 8));
        // A wave-task is WHOLE when its 256 samples lie inside the block's kept run and inside the window: true for
        // all but the first and last one or two of a slot, so the run of whole tasks [f_lo, f_hi) is cut out once per
        // slot and walked by a loop that tests nothing: one scalar add for the store's offset, one vector add for the
        // z window's address, the FIR, |.|, one store (round 5's loop spent 30 scalar instructions per task on the
        // three tests and the tap-count branch: SQ_INSTS_SALU 0.69 of SQ_INSTS_VALU).
        const int f_lo = max(wt_a, s_blk >= 0 ? 0 : (255 - s_blk) >> 8);
        const int f_hi = (int)min((int64_t)min(wt_b, tps >> 6), max((int64_t)0, (w_len - s_blk) >> 8));
        auto edge_task = [&](int wt) {
          const int k = wt * 64 + lane;
          if (k < tps) {
            const v2f* const zp = zs + (k >> lgi4);
            v2f are[2], aim[2];
            if (six) fir_taps<1, kT - 1>(zp, cf, are, aim);
            else fir_taps<0, kT>(zp, cf, are, aim);
            float res[4];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const v2f p2v = pk_fma(aim[pr], aim[pr], pk_mul(are[pr], are[pr]));
              res[2 * pr] = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2v.x) : p2v.x;
              res[2 * pr + 1] = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2v.y) : p2v.y;
            }
            // the window's edge (or the block's last samples) runs through this wave-task: one sample at a time,
            // each under its own test (four plain stores in a row may be merged into one 16-byte store by the
            // compiler, whose single range check would drop samples inside the window: synthp.hip)
            const int s0 = s_blk + 4 * k;                     // window-relative sample of res[0]
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if ((unsigned)(s0 + i) < (unsigned)w_len)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[i]), rsrc,
                                                      (unsigned)(s0 + i) * 4u, 0, GCWT_STORE_AUX);
          }
        };
        int wt = wt_a + ((wt0 - wt_a) & (kWavesI - 1));
        for (; wt < min(f_lo, wt_b); wt += kWavesI) edge_task(wt);
        if (wt < f_hi) {
          // z window of lane-task k = 64 wt + lane: element k >> lgi4, and a wave's tasks are kWavesI apart, so the
          // window moves by (64 kWavesI) >> lgi4 elements per task whatever the lane (I / 4 <= 256: static_assert above)
          const v2f* zp = zs + ((wt * 64 + lane) >> lgi4);
          const int z_step = (64 * kWavesI) >> lgi4;
          unsigned soff = (unsigned)(s_blk + 256 * wt) * 4u;             // scalar: byte offset of the task's first sample
          const unsigned voff = (unsigned)lane * 16u;
          typedef unsigned v4u __attribute__((ext_vector_type(4)));
          auto whole_task = [&](auto taps) {
            v2f are[2], aim[2];
            if constexpr (decltype(taps)::value == 6) fir_taps<1, kT - 1>(zp, cf, are, aim);
            else fir_taps<0, kT>(zp, cf, are, aim);
            v4u pk;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const v2f p2v = pk_fma(aim[pr], aim[pr], pk_mul(are[pr], are[pr]));
              pk[2 * pr] = __builtin_bit_cast(unsigned, MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2v.x) : p2v.x);
              pk[2 * pr + 1] = __builtin_bit_cast(unsigned, MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2v.y) : p2v.y);
            }
            __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, voff, soff, GCWT_STORE_AUX);
            zp += z_step;
            soff += 1024u * kWavesI;
          };
          if (six) for (; wt < f_hi; wt += kWavesI) whole_task(std::integral_constant<int, 6>());
          else for (; wt < f_hi; wt += kWavesI) whole_task(std::integral_constant<int, 8>());
        }
        for (; wt < wt_b; wt += kWavesI) edge_task(wt);
      }
    }
    __syncthreads();                        // z is read out before the next pass's exchange overwrites it
  }
}

hipError_t launch_synthi(int mode, const SynthiArgs& a, int n_items, int n_channels, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  if (mode != GCWT_OUT_AMPLITUDE_F32 && mode != GCWT_OUT_POWER_F32) return hipErrorInvalidValue;
  static bool attr_done[64] = {};            // per device: one process may drive several
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_done[dev_ & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synthi<GCWT_OUT_AMPLITUDE_F32>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synthi<GCWT_OUT_POWER_F32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if ((a.channels_fastest ? n_items : n_channels) > 65535) return hipErrorInvalidValue;
  const dim3 grid = a.channels_fastest ? dim3(n_channels, n_items) : dim3(n_items, n_channels), block(kThreadsI);
  // measure build only (option synthi_pad_kb): unused LDS on top of the kernel's own, i.e. fewer workgroups per CU --
  // how the launch time goes with occupancy (profiles/r04_store_study.md 5)
  int lds = kLdsBytes;
  if (kMeasureBuild && option_is_set("synthi_pad_kb")) {
    lds += 1024 * (int)option_or("synthi_pad_kb", 0);
    hipError_t e = hipFuncSetAttribute((const void*)k_synthi<GCWT_OUT_AMPLITUDE_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_synthi<GCWT_OUT_POWER_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
  }
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synthi<GCWT_OUT_AMPLITUDE_F32>), grid, block, lds, st, a);
  else
    hipLaunchKernelGGL((k_synthi<GCWT_OUT_POWER_F32>), grid, block, lds, st, a);
  return hipGetLastError();
}

}  // namespace gcwt
