// synth.hip -- the hot kernel of the engine: for every (channel, scale, block,
// phase) multiply the block spectrum by the scale's Morse filter and the polyphase
// twiddle, inverse-FFT 256 points, take |.| and store the samples in output order.
// (transforms.py:203-204: convolve each epoch with each scale's kernel, keep abs.)
// All three output modes; block layouts with halo > 32 use the 16-column kernel in
// kernels.hip.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "synth_math.h"

// cache policy bits of the output stores (measurement builds: profiles/r02_synth_study.md)
#ifndef GCWT_STORE_AUX
#define GCWT_STORE_AUX 2   // nt: the rows are written once and not read by this launch
#endif

namespace gcwt {

// ---------------------------------------------------------------------------
// k_synth7<MODE>: the production synthesis kernel.
//
// One workgroup owns a fixed set of 32 columns -- 32/R consecutive blocks x all R
// phases (R <= 32), or one block x 32 of its phases (R >= 64) -- and walks over
// the SCALES of the level.  What never changes for a thread,
//     P[k] = XB_blk[k] * W^{k r},   k = t + 16 j,
// is built once and kept in 32 VGPRs; what changes per batch, the scale's real gain
// G_s[k] (1 KB), is fetched eight scales at a time and parked in LDS, each lane's sixteen
// values side by side (four 16-byte reads per scale).  Per batch and thread:
//   v = P * G_s          complex x real, only the inputs below the scale's j_hi (the bins
//                        above its band are skipped: kernels.h, k_scale_windows), folded
//                        into the first radix-4 layer
//   rest of DFT16, W256 twiddle (table in LDS), transpose + re-deal through LDS
//   DFT16, |.|, 14 stores of 4 B per lane: 256 contiguous bytes per wave store, nt
// With 16 <= halo <= 32 rows 0 and 15 of a thread's 16 outputs are always halo,
// rows 2..13 are always kept and rows 1 / 14 are kept lane-wise.
// LDS: 16 x 513 complex + 256 complex + 8 x 320 gains + 256 indices + 256 half-sample factors
// = 79 KB -> two workgroups per CU.  128 VGPRs.
// ---------------------------------------------------------------------------
// exp(-2 pi i shift r / (256 R)): what phase r of a level whose band starts `shift` bins below zero
// carries (shift r < 2^24: exact in float before the division by a power of two)
__device__ __forceinline__ v2f phase_carrier(int shift, int r, int R) {
  float sn, cs;
  sincospif(-2.0f * (float)(shift * r) / (256.0f * (float)R), &sn, &cs);
  return (v2f){cs, sn};
}

// One column's sixteen first-stage outputs times W256^(t j) (tw: this lane's sixteen, two per 16-byte
// read; j = 0 is 1) into the exchange planes, STRIDE elements apart (0: run-time stride).
template <int STRIDE>
__device__ __forceinline__ void twiddle_to_planes(v2f* exw, const v2f (&v)[16], const v2f* tw, int stride = STRIDE) {
  const int st = STRIDE ? STRIDE : stride;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const v4f w2 = *reinterpret_cast<const v4f*>(tw + 2 * jj);
    exw[(2 * jj) * st] = jj == 0 ? v[0] : cmulv(v[dft16_pos(2 * jj)], (v2f){w2.x, w2.y});
    exw[(2 * jj + 1) * st] = cmulv(v[dft16_pos(2 * jj + 1)], (v2f){w2.z, w2.w});
  }
}

// WIDE: block halo above 48 (the long, heavy-tailed kernels of a level with a shifted band):
// every row's place in the block is tested, rows 0 and 15 included in the walk.
// Levels whose band starts band_shift bins below zero frequency (planner.h): bins are counted
// from the bottom of the band, so the sample the transform makes at block position m and
// phase r carries the carrier exp(-2 pi i shift (R (m_b + m) + r) / (256 R)) -- the block's
// constant goes into its spectrum, the phase's into P, and the position's into the W256
// twiddles and the order of the exchange planes (the same device as synthi.hip's
// demodulation): nothing is added to the scale loop.
template <int MODE, int NCOL, bool WIDE>
__global__ void __launch_bounds__(16 * NCOL, NCOL == 32 ? 4 : 3) k_synth7(const Synth7Args a) {
  constexpr int kThreads = 16 * NCOL;
  constexpr int kPlane = kThreads + 1;
  constexpr int kLgN = NCOL == 32 ? 5 : 4;
  // gains of one scale in LDS: lane t's sixteen (bins t + 16 j) side by side, 20 floats per lane so
  // that the four 16-byte reads of the 16 lanes of a column fall on distinct banks
  constexpr int kGainRow = 16 * 20;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);
  // W256 twiddles of the inter-stage multiply, lane t's sixteen side by side (pitch 18: 144 bytes, so
  // that they come in as 16-byte reads and the 16 lanes of a column fall on distinct banks)
  constexpr int kTwPitch = 18;
  v2f* const twl = ex + 16 * kPlane;
  float* const stage = reinterpret_cast<float*>(twl + 16 * kTwPitch);   // gains of kChunk scales
  int* const sc_lds = reinterpret_cast<int*>(stage + 8 * kGainRow);   // this level's scale indices
  v2f* const half_lds = reinterpret_cast<v2f*>(sc_lds + 256);         // the level's half-sample factors

  const Synth7Item it = a.items[blockIdx.x];
  const Synth7Level lv = a.levels[it.level];
  const int c = blockIdx.y;               // workspace slot: segment * n_channels + channel
  const int seg = c / a.seg.n_channels, ch = c - seg * a.seg.n_channels;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo, sh = lv.band_shift;
  {
    // The level grids are the union over the batch's segments: leave at once if this
    // group of blocks keeps no sample inside this segment's window (workgroup-uniform).
    const int64_t span = (int64_t)hop * R;
    const int64_t first = (int64_t)(lv.blk_base + it.blk0) * span;
    const int64_t last = first + (int64_t)(R > NCOL ? 1 : NCOL / R) * span;
    if (last <= a.seg.w_lo[seg] || first >= a.seg.w_hi[seg]) return;
  }
  const int tid = threadIdx.x;
  long long probe_c0 = 0, probe_t0 = 0, probe_ph[4] = {0, 0, 0, 0};
  if (kMeasureBuild && a.clock_probe) { probe_c0 = __builtin_amdgcn_s_memtime(); probe_t0 = __builtin_amdgcn_s_memrealtime(); }
  const int colw = tid >> 4, t = tid & 15;
  const bool wide = R > NCOL;
  const int blk_l = wide ? 0 : (colw >> lg);
  const int r = wide ? it.rtile * NCOL + colw : (colw & (R - 1));
  const int* const scales = a.scale_list + lv.scale_offset;
  constexpr int kChunk = 8;
  // ---- every global load of the prologue is issued here, before anything waits for one: the
  // workgroup pays one trip to memory, not one per table ----
  // this level's scale entries (read from LDS inside the loop: a global load there would have to
  // wait for vmcnt(0), i.e. for every store still in flight); at most 256 per level
  const int sc_v = tid < lv.n_scales ? scales[tid] : 0;
  // W256^((t - shift) j), parked as [t][j]
  const float2 tw_v = tid < 256 ? a.tw256[(((tid & 15) - sh) * (tid >> 4)) & 255] : make_float2(0.f, 0.f);
  // The filter enters as its real gain |H_s[k]|; the half-sample phase that even kernel
  // lengths carry is folded into P when the walk reaches those scales (they come last in
  // the level's list; its 256 factors wait in LDS).  Gains of kChunk scales at a time are
  // parked in LDS (10 KB), lane t's sixteen side by side; gain_lv holds them in that order
  // (k_scale_windows), one 16-byte load per thread and chunk, the next chunk's issued as soon as
  // the current one is parked.
  constexpr int kGainLoads = kChunk * 64 / kThreads;     // float4 per thread and chunk: 1 (2 for 16 columns)
  static_assert(kGainLoads * kThreads == kChunk * 64, "one chunk = a whole number of loads per thread");
  const float4* const gain_rows = reinterpret_cast<const float4*>(a.gain_lv + (int64_t)lv.scale_offset * 256);
  static_assert(kGainLoads == 1 || kGainLoads == 2, "one or two 16-byte loads per thread and chunk");
  // (two named registers, not an array: captured by the lambdas below an array of two went to scratch memory -- 48 bytes
  // of private segment per lane and a scratch set-up for every wave of the 16-column instantiation)
  float4 g_v0 = make_float4(0.f, 0.f, 0.f, 0.f), g_v1 = g_v0;
  auto load_gains = [&](int b0) {
    g_v0 = gain_rows[b0 * 64 + tid];
    if constexpr (kGainLoads > 1) g_v1 = gain_rows[b0 * 64 + kThreads + tid];
  };
  auto park_one = [&](int f, const float4& g) {          // float4 f of the chunk: scale f >> 6, lane (f >> 2) & 15
    *reinterpret_cast<float4*>(stage + (f >> 6) * kGainRow + ((f >> 2) & 15) * 20 + (f & 3) * 4) = g;
  };
  auto park_gains = [&]() {
    park_one(tid, g_v0);
    if constexpr (kGainLoads > 1) park_one(kThreads + tid, g_v1);
  };
  load_gains(0);
  const bool has_half = lv.n_plain < lv.n_scales;        // workgroup-uniform
  const float2 half_v = has_half && tid < 256 ? a.level_half_tw[lv.half_offset + tid] : make_float2(1.f, 0.f);
  const float2* ltw = a.level_tw + lv.tw_offset;
  const float2 b0 = ltw[t * r], st = ltw[16 * r];
  v2f pw[16];
  if (a.xr) {
    // Block spectra made here: XB_b = FFT_256(x_R[(b hop - halo + n) mod M]) / (256 P) for the
    // workgroup's 32/R blocks (one for R > 32), 16 threads per block, forward transform as
    // conj(IFFT(conj .)) on the packed inverse DFT16; the result goes through LDS to every
    // column (phase) of its block.  Saves the XB array's round trip through HBM and a launch
    // per level.  `ex` is free until the scale loop starts: [NCOL/2][16][16] exchange (element
    // (t, m2) at t*16 + (m2 ^ t): conflict-free without padding) + [NCOL/2][256] spectra
    // (at most NCOL/2 blocks per workgroup: R = 2).
    const int nblk_wg = wide ? 1 : (NCOL >> lg);
    v2f* const fx = ex;
    v2f* const xbs = ex + (NCOL / 2) * 256;
    static_assert(NCOL * 256 <= 16 * (16 * NCOL + 1), "prologue buffers must fit the exchange planes");
    v2f v[16];
    const int blkx = min(it.blk0 + colw, lv.nblk - 1);
    const int64_t m_b = (int64_t)(lv.blk_base + blkx) * hop - halo;       // the block's first decimated sample
    // the block's carrier: exp(-2 pi i shift m_b / 256)
    float2 cb = make_float2(1.f, 0.f);
    if (colw < nblk_wg) {
      if (sh) cb = a.tw256[(-(int64_t)sh * m_b) & 255];
      const float2* xr = a.xr + (int64_t)c * a.xr_cstride + lv.xr_offset;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 q = xr[(m_b + t + 16 * j) & lv.m_mask];
        v[j] = (v2f){q.x, -q.y};
      }
    }
    // the small tables first (they were asked for first), the samples stay in flight meanwhile
    if (tid < lv.n_scales) sc_lds[tid] = sc_v;
    if (tid < 256) { twl[(tid & 15) * kTwPitch + (tid >> 4)] = (v2f){tw_v.x, tw_v.y}; half_lds[tid] = (v2f){half_v.x, half_v.y}; }
    park_gains();
    if (lv.n_scales > kChunk) load_gains(kChunk);
    if (kMeasureBuild && a.clock_probe) { __builtin_amdgcn_s_waitcnt(0); probe_ph[0] = __builtin_amdgcn_s_memrealtime(); }
    if (colw < nblk_wg) {
      idft16v(v);
#pragma unroll
      for (int m = 0; m < 16; ++m) fx[colw * 256 + t * 16 + (m ^ t)] = v[dft16_pos(m)];
    }
    __syncthreads();
    if (kMeasureBuild && a.clock_probe) probe_ph[1] = __builtin_amdgcn_s_memrealtime();
    if (colw < nblk_wg) {
      // element (writer k1, index t) arrives without its twiddle exp(+2 pi i k1 t / 256): with no
      // band shift that is twl[t][k1], just parked
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = fx[colw * 256 + k1 * 16 + (t ^ k1)];
      if (sh) {                              // workgroup-uniform
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
          const float2 w = a.tw256[(t * k1) & 255];
          v[k1] = cmulv(v[k1], (v2f){w.x, w.y});
        }
      } else {
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmulv(v[k1], twl[t * kTwPitch + k1]);
      }
      idft16v(v);
      const float xs = a.xb_scale;
      if (sh) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const v2f z = v[dft16_pos(j)];
          xbs[colw * 256 + t + 16 * j] = cmulv((v2f){z.x * xs, -z.y * xs}, (v2f){cb.x, cb.y});
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const v2f z = v[dft16_pos(j)];
          xbs[colw * 256 + t + 16 * j] = (v2f){z.x * xs, -z.y * xs};
        }
      }
    }
    __syncthreads();
    if (kMeasureBuild && a.clock_probe) probe_ph[2] = __builtin_amdgcn_s_memrealtime();
    v2f wcur = (v2f){b0.x, b0.y};
    if (sh) wcur = cmulv(wcur, phase_carrier(sh, r, R));
    const v2f wstep = (v2f){st.x, st.y};
    const v2f* const mine = xbs + blk_l * 256 + t;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      pw[j] = cmulv(mine[16 * j], wcur);
      wcur = cmulv(wcur, wstep);
    }
  } else {
    // it.blk0 counts from the level's first computed block (lv.blk_base); columns past
    // the last block reuse it and are never stored
    const int blk = min(it.blk0 + blk_l, lv.nblk - 1);
    const float2* xb = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset + (int64_t)blk * 256 + t;
    v2f wcur = (v2f){b0.x, b0.y};
    if (sh) {   // the phase's and the block's carriers (the XB pass knows nothing of the shift)
      const float2 cb = a.tw256[(-(int64_t)sh * ((int64_t)(lv.blk_base + blk) * hop - halo)) & 255];
      wcur = cmulv(cmulv(wcur, phase_carrier(sh, r, R)), (v2f){cb.x, cb.y});
    }
    const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 q = xb[16 * j];
      pw[j] = cmulv((v2f){q.x, q.y}, wcur);
      wcur = cmulv(wcur, wstep);
    }
    if (tid < lv.n_scales) sc_lds[tid] = sc_v;
    if (tid < 256) { twl[(tid & 15) * kTwPitch + (tid >> 4)] = (v2f){tw_v.x, tw_v.y}; half_lds[tid] = (v2f){half_v.x, half_v.y}; }
    park_gains();
    if (lv.n_scales > kChunk) load_gains(kChunk);
  }
  const int sstride = wide ? NCOL : R;
  v2f* const exw = ex + ((t - sh) & 15) * kPlane + (wide ? colw : (blk_l << (4 + lg)) + r);
  const int blk_l2 = wide ? 0 : (tid >> (4 + lg));
  const int rem = wide ? tid : (tid & ((16 << lg) - 1));
  const int m2 = wide ? (tid >> kLgN) : (rem >> lg);
  const int r2 = wide ? it.rtile * NCOL + (tid & (NCOL - 1)) : (rem & (R - 1));
  const v2f* const exr = ex + tid;
  const int off0 = (blk_l2 * hop + m2 - halo) * R + r2;   // sample offset of row m1 = 0
  const int m1step = 16 * R;
  // which of a thread's 16 output rows (m = 16 m1 + m2) lie in the kept part [halo, 256 - halo):
  // 16 <= halo <= 32: rows 2..13 always, 1 and 14 lane-wise; 32 < halo <= 48 (long kernels at
  // the decimation cap, narrow-band wavelets): rows 3..12 always, 2 and 13 lane-wise
  const bool deep = halo > 32;
  const bool keep1 = !deep && m2 >= halo - 16, keep14 = !deep && m2 < 32 - halo;
  const bool keep2 = !deep || m2 >= halo - 32, keep13 = !deep || m2 < 48 - halo;
  // WIDE (halo > 48): row m1 is kept when halo <= 16 m1 + m2 < 256 - halo: bit m1 of `keep_rows`
  unsigned keep_rows = 0;
  if (WIDE) {
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) keep_rows |= (16 * m1 + m2 >= halo && 16 * m1 + m2 < 256 - halo) ? 1u << m1 : 0u;
  }

  // Stores go through a buffer descriptor that covers exactly the samples this launch
  // may write, [w_lo, w_hi) of the segment: anything else -- halo rows (negative offsets
  // wrap), the end of the epoch, blocks past the last one, neighbouring time blocks -- is
  // dropped by the hardware range check, so the store loop carries no bound tests.
  constexpr int kElem = MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1;   // floats per output sample
  const int64_t n_b = (int64_t)(lv.blk_base + it.blk0) * hop * R;   // first sample of the block group
  const int64_t w_lo = a.seg.w_lo[seg];
  const int64_t w_len = a.seg.w_hi[seg] - w_lo;
  const unsigned ext_bytes = w_len > 0 && !(kMeasureBuild && (a.drop_stores & 1)) ? (unsigned)(w_len * (4 * kElem)) : 0u;
  float* const out0 = a.out + ((int64_t)ch * a.n_scales * a.row_len + a.seg.seg_col[seg] + w_lo) * kElem;
  const unsigned voff0 = (unsigned)(((int)(n_b - w_lo) + off0) * (4 * kElem));
  const unsigned vstep = (unsigned)(m1step * (4 * kElem));
  // R = 2 (amplitude / power): rows leave in pairs, see the store loop
  const bool pair_rows = !wide && lg == 1;
  const bool upper = (tid & 32) != 0;
  const bool keep_p0 = upper ? keep2 : keep1, keep_p6 = upper ? keep14 : keep13;
  const unsigned voff_pair = voff0 - (upper ? (unsigned)((2 * hop - 32) * 4) : 0u);   // the wave's first block, lane-linear
  const unsigned pair_step = (unsigned)(2 * hop * 4);                              // its second block
  const float* const st_rd = stage + t * 20;
  __syncthreads();
  if (kMeasureBuild && a.clock_probe) probe_ph[3] = __builtin_amdgcn_s_memrealtime();

  for (int b = 0; b < lv.n_scales; ++b) {
    if (b > 0 && (b & (kChunk - 1)) == 0) {    // wave-uniform
      __syncthreads();                         // everyone is done with the previous chunk
      park_gains();                            // asked for eight scales ago
      if (b + kChunk < lv.n_scales) load_gains(b + kChunk);
      __syncthreads();
    }
    if (b == lv.n_plain) {                     // wave-uniform; at most once per workgroup
#pragma unroll
      for (int j = 0; j < 16; ++j) pw[j] = cmulv(pw[j], half_lds[t + 16 * j]);
    }
    const float4* const hs = reinterpret_cast<const float4*>(st_rd + (b & (kChunk - 1)) * kGainRow);
    // the entry's top byte: 16 - j_hi, first-pass inputs j >= j_hi are left out for this scale (kernels.h)
    const int entry = __builtin_amdgcn_readfirstlane(sc_lds[b]);
    v2f v[16];
    switch ((unsigned)entry >> 24) {             // wave-uniform; 16 - j_hi
#define GCWT_WINDOW(hi) case 16 - (hi): gain_first_layer<hi>(v, pw, hs); break;
      GCWT_WINDOW(15) GCWT_WINDOW(14) GCWT_WINDOW(13) GCWT_WINDOW(12) GCWT_WINDOW(11) GCWT_WINDOW(10) GCWT_WINDOW(9)
#undef GCWT_WINDOW
      default: gain_first_layer<16>(v, pw, hs); break;
    }
    idft16v_tail(v);
    // the column's sixteen values, W256^(t j) applied, into its exchange planes: with the stride between
    // them known at compile time the stores need no address arithmetic and pair up (ds_write2_b64)
    switch (sstride) {                         // workgroup-uniform: R, or the column count for R > columns
      case 2: twiddle_to_planes<2>(exw, v, twl + t * kTwPitch); break;
      case 4: twiddle_to_planes<4>(exw, v, twl + t * kTwPitch); break;
      case 8: twiddle_to_planes<8>(exw, v, twl + t * kTwPitch); break;
      case 16: twiddle_to_planes<16>(exw, v, twl + t * kTwPitch); break;
      case 32: twiddle_to_planes<32>(exw, v, twl + t * kTwPitch); break;
      default: twiddle_to_planes<0>(exw, v, twl + t * kTwPitch, sstride); break;
    }
    if (!(kMeasureBuild && (a.drop_stores & 2))) __syncthreads();      // (always taken in the product build: kernels.h)
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = exr[k1 * kPlane];
    if (!(kMeasureBuild && (a.drop_stores & 2))) __syncthreads();
    idft16v(v);

    // descriptor built from provably wave-uniform words (else hipcc waterfalls every store)
    const int srow = entry & kScaleIndexMask;
    const uint64_t dst_bits = reinterpret_cast<uint64_t>(out0 + (int64_t)srow * a.row_len * kElem);
    const uint32_t dst_lo = __builtin_amdgcn_readfirstlane((uint32_t)dst_bits);
    const uint32_t dst_hi = __builtin_amdgcn_readfirstlane((uint32_t)(dst_bits >> 32));
    float* const dst = reinterpret_cast<float*>(((uint64_t)dst_hi << 32) | dst_lo);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        dst, 0, __builtin_amdgcn_readfirstlane(ext_bytes), 0x00020000);
    if (MODE != GCWT_OUT_COMPLEX_C64 && !WIDE && pair_rows) {      // R = 2; workgroup-uniform
      // A wave holds two blocks x 32 samples of every row: 128-byte pieces.  Swapping the upper
      // half of row m1 with the lower half of row m1 + 1 (v_permlane32_swap) leaves one register
      // with 64 consecutive samples of the first block and one with the second block's: 256
      // contiguous bytes per store like every other level.
#pragma unroll
      for (int m1 = 1; m1 < 15; m1 += 2) {
        const v2f z0 = v[dft16_pos(m1)], z1 = v[dft16_pos(m1 + 1)];
        const float p0 = __builtin_fmaf(z0.y, z0.y, z0.x * z0.x), p1 = __builtin_fmaf(z1.y, z1.y, z1.x * z1.x);
        const float a0 = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p0) : p0;
        const float a1 = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p1) : p1;
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0),
                                                         __builtin_bit_cast(unsigned, a1), false, false);
        const bool keep = m1 == 1 ? keep_p0 : m1 == 13 ? keep_p6 : true;   // this lane's row: m1 + (lane >= 32)
        const unsigned vo = voff_pair + (unsigned)m1 * vstep;
        if (keep) {
          __builtin_amdgcn_raw_buffer_store_b32(sw[0], rsrc, vo, 0, GCWT_STORE_AUX);
          __builtin_amdgcn_raw_buffer_store_b32(sw[1], rsrc, vo + pair_step, 0, GCWT_STORE_AUX);
        }
      }
      continue;
    }
#pragma unroll
    for (int m1 = 1; m1 < 15; ++m1) {     // (rows 0 and 15 are never kept: halo >= 16)
      const v2f z = v[dft16_pos(m1)];
      const bool keep = WIDE ? (keep_rows >> m1) & 1u
                             : m1 == 1 ? keep1 : m1 == 2 ? keep2 : m1 == 13 ? keep13 : m1 == 14 ? keep14 : true;
      const unsigned vo = voff0 + (unsigned)m1 * vstep;
      if (MODE == GCWT_OUT_COMPLEX_C64) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        if (keep) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, z), rsrc, vo, 0, GCWT_STORE_AUX);
      } else {
        const float p2 = __builtin_fmaf(z.y, z.y, z.x * z.x);   // v_mul + v_fma: cheaper than packed
        const float val = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
        if (keep) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc, vo, 0, GCWT_STORE_AUX);
      }
    }
  }
  if (kMeasureBuild && a.clock_probe && tid == 0) {
    atomicAdd(a.clock_probe, (unsigned long long)(__builtin_amdgcn_s_memtime() - probe_c0));
    atomicAdd(a.clock_probe + 1, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - probe_t0));
    for (int i = 0; i < 4; ++i) atomicAdd(a.clock_probe + 2 + i, (unsigned long long)(probe_ph[i] - probe_t0));
    atomicAdd(a.clock_probe + 6, 1ull);
  }
}

template <int NCOL, bool WIDE>
static hipError_t launch_synth7_n(int mode, const Synth7Args& a, int n_items, int n_channels,
                                  hipStream_t st) {
  constexpr int lds = 16 * (16 * NCOL + 1) * 8 + 16 * 18 * 8 + 8 * 320 * 4 + 256 * 4 + 256 * 8;
  static bool attr_done[64] = {};            // per device: one process may drive several
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_done[dev_ & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synth7<GCWT_OUT_AMPLITUDE_F32, NCOL, WIDE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth7<GCWT_OUT_POWER_F32, NCOL, WIDE>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth7<GCWT_OUT_COMPLEX_C64, NCOL, WIDE>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(16 * NCOL);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth7<GCWT_OUT_AMPLITUDE_F32, NCOL, WIDE>), grid, block, lds, st, a);
  else if (mode == GCWT_OUT_POWER_F32)
    hipLaunchKernelGGL((k_synth7<GCWT_OUT_POWER_F32, NCOL, WIDE>), grid, block, lds, st, a);
  else
    hipLaunchKernelGGL((k_synth7<GCWT_OUT_COMPLEX_C64, NCOL, WIDE>), grid, block, lds, st, a);
  return hipGetLastError();
}

hipError_t launch_synth7(int mode, int ncol, bool wide_halo, const Synth7Args& a, int n_items, int n_channels,
                         hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  if (wide_halo)
    return ncol == 16 ? launch_synth7_n<16, true>(mode, a, n_items, n_channels, st)
                      : launch_synth7_n<32, true>(mode, a, n_items, n_channels, st);
  return ncol == 16 ? launch_synth7_n<16, false>(mode, a, n_items, n_channels, st)
                    : launch_synth7_n<32, false>(mode, a, n_items, n_channels, st);
}

}  // namespace gcwt
