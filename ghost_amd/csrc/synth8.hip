// synth8.hip -- k_synth8: the synthesis of synth.hip's k_synth7 (same arithmetic, same
// results, same column layout and store pattern) with the two halves of each 256-point
// inverse FFT given to DIFFERENT waves of one 1024-thread workgroup:
//
//   producers (waves 0-7)   filter gain x P[k], DFT16, W256 twiddle, transposing LDS writes
//   consumers (waves 8-15)  LDS reads, DFT16, |.| / |.|^2 / complex, stores
//
// through a double-buffered exchange: while the consumers take scale b out of one buffer
// the producers fill the other with scale b+1; one workgroup barrier per scale.
//
// Why: on gfx950 the LDS pipe and the vector ALU work concurrently only when different
// waves feed them (tools/role_overlap.hip: 1434 + 929 cycles side by side vs 1423 / 836
// alone), while waves that all run the same pass-1 / exchange / pass-2 sequence fall into
// step and pay VALU time PLUS LDS time (tools/valu_lds_overlap.hip: 2435 + 3521 -> 5529).
// Splitting the roles also frees registers: a producer keeps its 16 W256 twiddles in
// VGPRs for the whole kernel (k_synth7 re-reads them from LDS for every scale, each read
// followed by a wait), and its global loads (the next scales' gains) never queue behind
// the consumers' stores.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "synth_math.h"

namespace gcwt {

// NCOL = 32: one 1024-thread workgroup per CU; NCOL = 16: 512 threads, two per CU (75 KB of
// LDS each), whose prologues and barriers then overlap each other's steady state.
template <int NCOL>
struct Synth8Cfg {
  static constexpr int kCols = NCOL;                 // columns per workgroup
  static constexpr int kRole = 16 * NCOL;            // threads per role
  static constexpr int kPlane8 = kRole + 1;          // v2f elements per exchange plane
  static constexpr int kChunk8 = NCOL / 4;           // scales whose gains are staged in LDS at a time
  static constexpr int kLgChunk = NCOL == 32 ? 3 : 2;
  static constexpr int kLgN = NCOL == 32 ? 5 : 4;
  static constexpr int kLds8 = 2 * 16 * kPlane8 * 8 + 2 * kChunk8 * 256 * 4 + 256 * 4;
};

template <int MODE, int NCOL>
__global__ void __launch_bounds__(32 * NCOL) k_synth8(const Synth7Args a) {
  constexpr int kCols = Synth8Cfg<NCOL>::kCols, kRole = Synth8Cfg<NCOL>::kRole;
  constexpr int kPlane8 = Synth8Cfg<NCOL>::kPlane8, kChunk8 = Synth8Cfg<NCOL>::kChunk8;
  constexpr int kLgChunk = Synth8Cfg<NCOL>::kLgChunk, kLgN = Synth8Cfg<NCOL>::kLgN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);                       // [2][16][kPlane8]
  float* const stage = reinterpret_cast<float*>(ex + 2 * 16 * kPlane8);  // [2][kChunk8][256]
  int* const sc_lds = reinterpret_cast<int*>(stage + 2 * kChunk8 * 256);

  const Synth7Item it = a.items[blockIdx.x];
  const Synth7Level lv = a.levels[it.level];
  const int c = blockIdx.y;               // workspace slot: segment * n_channels + channel
  const int seg = c / a.seg.n_channels, ch = c - seg * a.seg.n_channels;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo;
  {
    // union grids of a batch: leave at once if this group of blocks keeps no sample
    // inside this segment's window (workgroup-uniform)
    const int64_t span = (int64_t)hop * R;
    const int64_t first = (int64_t)(lv.blk_base + it.blk0) * span;
    const int64_t last = first + (int64_t)(R > kCols ? 1 : kCols / R) * span;
    if (last <= a.seg.w_lo[seg] || first >= a.seg.w_hi[seg]) return;
  }
  long long probe_c0 = 0, probe_t0 = 0;
  if (a.clock_probe) { probe_c0 = __builtin_amdgcn_s_memtime(); probe_t0 = __builtin_amdgcn_s_memrealtime(); }
  const bool producer = threadIdx.x < kRole;            // wave-uniform
  const int tid = producer ? threadIdx.x : threadIdx.x - kRole;
  const bool wide = R > kCols;
  const int n_scales = lv.n_scales;
  const int* const scales = a.scale_list + lv.scale_offset;
  for (int i = threadIdx.x; i < n_scales; i += 2 * kRole) sc_lds[i] = scales[i] & kScaleIndexMask;   // (this kernel computes every first-pass input)

  // The two roles run separate loops (their registers never coexist); the workgroup barrier
  // counts waves, not call sites, and both loops pass it exactly 2 + n_scales times.
  if (producer) {
    const int colw = tid >> 4, t = tid & 15;
    float gq[4];                            // gains of the next chunk on their way to LDS
    auto load_gains = [&](int b0) {         // chunk starting at scale b0 -> registers
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = tid + kRole * q;
        const int sb = min(b0 + (i >> 8), n_scales - 1);
        gq[q] = a.gain[(int64_t)(scales[sb] & kScaleIndexMask) * 256 + (i & 255)];
      }
    };
    auto park_gains = [&](int chunk) {      // registers -> stage[chunk & 1]
      float* const dst = stage + (chunk & 1) * (kChunk8 * 256);
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[tid + kRole * q] = gq[q];
    };
    load_gains(0);
    // W256^(t j): constant for the thread, kept in registers for every scale
    v2f pw[16], tw[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 w = a.tw256[(t * j) & 255];
      tw[j] = (v2f){w.x, w.y};
    }
    // P[k] = XB_blk[k] * W^{k r}, k = t + 16 j; columns past the last block reuse it and
    // are never stored
    const int blk_l = wide ? 0 : (colw >> lg);
    const int r = wide ? it.rtile * kCols + colw : (colw & (R - 1));
    {
      const int blk = min(it.blk0 + blk_l, lv.nblk - 1);
      const float2* xb = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset + (int64_t)blk * 256 + t;
      const float2* ltw = a.level_tw + lv.tw_offset;
      const float2 b0 = ltw[t * r], st = ltw[16 * r];
      v2f wcur = (v2f){b0.x, b0.y};
      const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 q = xb[16 * j];
        pw[j] = cmulv((v2f){q.x, q.y}, wcur);
        wcur = cmulv(wcur, wstep);
      }
    }
    const int sstride = wide ? kCols : R;
    v2f* const exw = ex + t * kPlane8 + (wide ? colw : (blk_l << (4 + lg)) + r);
    park_gains(0);
    __syncthreads();                        // scale list and the first gains are in LDS

    for (int nb = 0; nb < n_scales; ++nb) { // scale nb -> exchange buffer nb & 1
      const int next0 = ((nb >> kLgChunk) + 1) * kChunk8;       // first scale of the chunk after nb's
      if ((nb & (kChunk8 - 1)) == kChunk8 - 2 && next0 < n_scales) load_gains(next0);
      if (nb == lv.n_plain) {               // wave-uniform; at most once: the even kernel lengths
        const float2* hp = a.level_half_tw + lv.half_offset + t;   // carry a half-sample phase
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float2 q = hp[16 * j];
          pw[j] = cmulv(pw[j], (v2f){q.x, q.y});
        }
      }
      const float* const hs = stage + ((nb >> kLgChunk) & 1) * (kChunk8 * 256) + (nb & (kChunk8 - 1)) * 256 + t;
      v2f v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = pw[j] * hs[16 * j];
      idft16v(v);
      v2f* const w = exw + (nb & 1) * (16 * kPlane8);
#pragma unroll
      for (int j = 0; j < 16; ++j) w[j * sstride] = cmulv(v[dft16_pos(j)], tw[j]);
      if ((nb & (kChunk8 - 1)) == kChunk8 - 1 && next0 < n_scales) park_gains(next0 >> kLgChunk);
      __syncthreads();                      // scale nb in place (and scale nb - 1 taken out)
    }
    __syncthreads();                        // the consumers' last scale
  } else {
    constexpr int kElem = MODE == GCWT_OUT_COMPLEX_C64 ? 2 : 1;   // floats per output sample
    // The transpose re-deals columns over lanes so that consecutive lanes hold consecutive
    // output samples: each wave store writes 256 contiguous bytes straight from registers.
    const int blk_l2 = wide ? 0 : (tid >> (4 + lg));
    const int rem = wide ? tid : (tid & ((16 << lg) - 1));
    const int m2 = wide ? (tid >> kLgN) : (rem >> lg);
    const int r2 = wide ? it.rtile * kCols + (tid & (kCols - 1)) : (rem & (R - 1));
    const v2f* const exr = ex + tid;
    const int off0 = (blk_l2 * hop + m2 - halo) * R + r2;   // sample offset of row m1 = 0
    // which of a thread's 16 output rows (m = 16 m1 + m2) lie in the kept part [halo, 256 - halo):
    // 16 <= halo <= 32: rows 2..13 always, 1 and 14 lane-wise; 32 < halo <= 48: rows 3..12
    // always, 2 and 13 lane-wise
    const bool deep = halo > 32;
    const bool keep1 = !deep && m2 >= halo - 16, keep14 = !deep && m2 < 32 - halo;
    const bool keep2 = !deep || m2 >= halo - 32, keep13 = !deep || m2 < 48 - halo;
    // Stores go through a buffer descriptor that covers exactly the samples this launch may
    // write, [w_lo, w_hi) of the segment: halo rows (negative offsets wrap), the end of the
    // epoch, blocks past the last one, neighbouring time blocks are dropped by the hardware
    // range check, so the store loop carries no bound tests.
    const int64_t n_b = (int64_t)(lv.blk_base + it.blk0) * hop * R;   // first sample of the block group
    const int64_t w_lo = a.seg.w_lo[seg];
    const int64_t w_len = a.seg.w_hi[seg] - w_lo;
    const unsigned ext_bytes = w_len > 0 && !(a.drop_stores & 1) ? (unsigned)(w_len * (4 * kElem)) : 0u;
    float* const out0 = a.out + ((int64_t)ch * a.n_scales * a.row_len + a.seg.seg_col[seg] + w_lo) * kElem;
    const unsigned voff0 = (unsigned)(((int)(n_b - w_lo) + off0) * (4 * kElem));
    const unsigned vstep = (unsigned)(16 * R * (4 * kElem));
    __syncthreads();                        // (tables)
    __syncthreads();                        // scale 0 in place

    for (int b = 0; b < n_scales; ++b) {
      const v2f* const rd = exr + (b & 1) * (16 * kPlane8);
      v2f v[16];
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = rd[k1 * kPlane8];
      idft16v(v);
      // descriptor built from provably wave-uniform words (else hipcc waterfalls every store)
      const int srow = __builtin_amdgcn_readfirstlane(sc_lds[b]);
      const uint64_t dst_bits = reinterpret_cast<uint64_t>(out0 + (int64_t)srow * a.row_len * kElem);
      const uint32_t dst_lo = __builtin_amdgcn_readfirstlane((uint32_t)dst_bits);
      const uint32_t dst_hi = __builtin_amdgcn_readfirstlane((uint32_t)(dst_bits >> 32));
      float* const dst = reinterpret_cast<float*>(((uint64_t)dst_hi << 32) | dst_lo);
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          dst, 0, __builtin_amdgcn_readfirstlane(ext_bytes), 0x00020000);
      // 16 <= halo <= 32: rows 0 and 15 are always halo, 2..13 always kept, 1 / 14 lane-wise
#pragma unroll
      for (int m1 = 1; m1 < 15; ++m1) {
        const v2f z = v[dft16_pos(m1)];
        const bool keep = m1 == 1 ? keep1 : m1 == 2 ? keep2 : m1 == 13 ? keep13 : m1 == 14 ? keep14 : true;
        const unsigned vo = voff0 + (unsigned)m1 * vstep;
        if (MODE == GCWT_OUT_COMPLEX_C64) {
          typedef unsigned v2u __attribute__((ext_vector_type(2)));
          if (keep) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, z), rsrc, vo, 0, 0);
        } else {
          const float p2 = __builtin_fmaf(z.y, z.y, z.x * z.x);
          const float val = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
          if (keep) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc, vo, 0, 0);
        }
      }
      __syncthreads();                      // scale b taken out (and scale b + 1 in place)
    }
    if (a.clock_probe && tid == 0) {
      atomicAdd(a.clock_probe, (unsigned long long)(__builtin_amdgcn_s_memtime() - probe_c0));
      atomicAdd(a.clock_probe + 1, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - probe_t0));
    }
  }
}

template <int NCOL>
static hipError_t launch_synth8_n(int mode, const Synth7Args& a, int n_items, int n_channels, hipStream_t st) {
  constexpr int lds = Synth8Cfg<NCOL>::kLds8;
  static bool attr_done[64] = {};            // per device: one process may drive several
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_done[dev_ & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synth8<GCWT_OUT_AMPLITUDE_F32, NCOL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth8<GCWT_OUT_POWER_F32, NCOL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth8<GCWT_OUT_COMPLEX_C64, NCOL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(32 * NCOL);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth8<GCWT_OUT_AMPLITUDE_F32, NCOL>), grid, block, lds, st, a);
  else if (mode == GCWT_OUT_POWER_F32)
    hipLaunchKernelGGL((k_synth8<GCWT_OUT_POWER_F32, NCOL>), grid, block, lds, st, a);
  else
    hipLaunchKernelGGL((k_synth8<GCWT_OUT_COMPLEX_C64, NCOL>), grid, block, lds, st, a);
  return hipGetLastError();
}

hipError_t launch_synth8(int mode, int ncol, const Synth7Args& a, int n_items, int n_channels,
                         hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  return ncol == 16 ? launch_synth8_n<16>(mode, a, n_items, n_channels, st)
                    : launch_synth8_n<32>(mode, a, n_items, n_channels, st);
}

}  // namespace gcwt
