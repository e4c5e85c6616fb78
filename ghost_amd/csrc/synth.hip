// synth.hip -- the hot kernel: fused filter * polyphase twiddle * 256-point
// inverse FFT * |.| * coalesced store, for amplitude / power output.
// (transforms.py:203-204: convolve each epoch with each scale's kernel and keep abs.)
//
// One workgroup (512 threads = 8 waves) owns one (channel, scale, column tile,
// block range).  A "column" is one 256-point inverse FFT: (block b, phase r),
//   y[R (b*hop + m - halo) + r] = sum_k XB_b[k] H_s[k] W^{k r} e^{2 pi i k m/256},
//   W = e^{2 pi i/(256 R)}.
// 32 columns are in flight per batch, 16 threads per column, 16 points per thread:
//   lane = (column mod 4) * 16 + t,  wave w holds columns 4w .. 4w+3.
// * HW[k] = H_s[k] W^{k r} is fixed for a thread (its column's r never changes),
//   so it lives in registers for the whole workgroup: one complex multiply per
//   input point.
// * 256 = 16 x 16: DFT16 in registers, W256 twiddle, 16x16 transpose through LDS
//   between the 16 lanes of a column (same wave: no workgroup barrier), DFT16.
// * amplitudes go to an LDS tile laid out like the output row, XOR-swizzled by
//   16-byte groups so the column-wise writes spread over the banks, and leave as
//   whole 128-byte lines; the tile is double buffered so the only workgroup
//   barrier is one per batch and the stores overlap the next batch's arithmetic.
// * next batch's block spectra are prefetched into registers before the FFT.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gcwt {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Packed complex arithmetic on (re, im) register pairs.  The operand swizzles and
// sign flips ride on the VOP3P op_sel / neg modifiers, so a complex multiply is two
// instructions and a multiply by +-i is free (hipcc does not fold these itself).
//   lo result uses S[op_sel], hi result uses S[op_sel_hi]; neg_lo / neg_hi likewise.
__device__ __forceinline__ v2f cmulv(v2f a, v2f w) {   // a * w
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
__device__ __forceinline__ v2f add_ib(v2f a, v2f b) {  // a + i b
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f sub_ib(v2f a, v2f b) {  // a - i b
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// radix-4 butterfly for the inverse transform (W4 = +i); c_times_i: c enters as i*c
template <bool C_TIMES_I>
__device__ __forceinline__ void bfly4(v2f& a, v2f& b, v2f& c, v2f& d) {
  const v2f s0 = C_TIMES_I ? add_ib(a, c) : a + c;
  const v2f s1 = C_TIMES_I ? sub_ib(a, c) : a - c;
  const v2f s2 = b + d, u = b - d;
  a = s0 + s2;
  c = s0 - s2;
  b = add_ib(s1, u);
  d = sub_ib(s1, u);
}

// 16-point inverse DFT in registers (exp(+2 pi i n k/16)).  Input natural order;
// output X[4 k1 + k2] is left in v[k1 + 4 k2] (use dft16_pos to address it).
__host__ __device__ constexpr int dft16_pos(int k) { return (k >> 2) | ((k & 3) << 2); }

__device__ __forceinline__ void idft16v(v2f v[16]) {
  const v2f w1 = {0.92387953251128674f, 0.38268343236508977f};   // W16^1
  const v2f w3 = {0.38268343236508977f, 0.92387953251128674f};   // W16^3
  const v2f w9 = {-0.92387953251128674f, -0.38268343236508977f}; // W16^9
  const float h = 0.70710678118654752f;
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) bfly4<false>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
  // v[n1 + 4 k2] *= W16^(n1 k2); W16^2 = h(1+i), W16^6 = h(-1+i); W16^4 = i is folded below
  v[5] = cmulv(v[5], w1);
  v[9] = add_ib(v[9], v[9]) * h;
  v[13] = cmulv(v[13], w3);
  v[6] = add_ib(v[6], v[6]) * h;
  v[14] = sub_ib(v[14], v[14]) * (-h);
  v[7] = cmulv(v[7], w3);
  v[11] = sub_ib(v[11], v[11]) * (-h);
  v[15] = cmulv(v[15], w9);
  bfly4<false>(v[0], v[1], v[2], v[3]);
  bfly4<false>(v[4], v[5], v[6], v[7]);
  bfly4<true>(v[8], v[9], v[10], v[11]);
  bfly4<false>(v[12], v[13], v[14], v[15]);
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kS2Cols = 32;                   // columns per batch
constexpr int kS2ExCol = 16 * 17;             // v2f per column in the exchange area
constexpr int kS2ExBytes = kS2Cols * kS2ExCol * 8;
constexpr int kS2TileFloats = kS2Cols * 256;  // one tile buffer
constexpr int kS2LdsBytes = kS2ExBytes + 2 * kS2TileFloats * 4;

template <int MODE>
__global__ void __launch_bounds__(512) k_synth2(const Synth2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex_all = reinterpret_cast<v2f*>(smem);
  float* const tiles = reinterpret_cast<float*>(smem + kS2ExBytes);

  const Synth2Item it = a.items[blockIdx.x];
  const Synth2Level lv = a.levels[it.level];
  const int c = blockIdx.y;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo;
  const int tid = threadIdx.x;
  const int colw = tid >> 4, t = tid & 15;     // column within the batch, thread within the column
  const bool wide = R > 32;                    // one block per batch, 32 of its R phases
  const int bpb = wide ? 1 : (32 >> lg);       // blocks per batch
  const int blk_l = wide ? 0 : (colw >> lg);
  const int r = wide ? it.rtile * 32 + colw : (colw & (R - 1));

  // -- per-thread constants -------------------------------------------------
  v2f tw[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float2 w = a.tw256[(t * j) & 255];
    tw[j] = (v2f){w.x, w.y};
  }
  v2f hw[16];
  {
    const float2* bank = a.bank + (int64_t)it.scale * 256 + t;
    const float2* ltw = a.level_tw + lv.tw_offset;
    const float2 b0 = ltw[t * r], st = ltw[16 * r];
    v2f wcur = (v2f){b0.x, b0.y};
    const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 hk = bank[16 * j];
      hw[j] = cmulv((v2f){hk.x, hk.y}, wcur);
      wcur = cmulv(wcur, wstep);
    }
  }
  v2f* const ex = ex_all + colw * kS2ExCol;
  // tile position of (m = t + 16 m1): wbase + mstride * m1, 16-byte groups swizzled by s
  const int rowlen = wide ? 32 : R;
  const int sbits = rowlen >= 8 ? (rowlen >> 2) - 1 : 0;
  const int swz = rowlen >= 8 ? ((t >> (rowlen >= 32 ? 0 : (5 - lg))) & sbits) : 0;
  const int wbase = (wide ? (t * 32 + colw) : ((blk_l * 256 + t) * R + r)) ^ (swz << 2);
  const int mstride = 16 * rowlen;

  const float2* xb = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset + t;
  float* const outrow = a.out + ((int64_t)c * a.n_scales + it.scale) * a.n_samples + a.epoch_start;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(outrow) & 15) == 0;

  // -- first batch's spectra ---------------------------------------------------
  v2f xn[16];
  {
    // columns past the last block read block nblk-1 again; their output is never stored
    const int blk = min(it.blk0 + blk_l, lv.nblk - 1);
    const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 q = p[16 * j];
      xn[j] = (v2f){q.x, q.y};
    }
  }

  for (int b = 0; b < it.nbatch; ++b) {
    const int blk0 = it.blk0 + b * bpb;        // first block of this batch
    v2f v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmulv(xn[j], hw[j]);
    if (b + 1 < it.nbatch && !(a.pad & 4)) {   // prefetch the next batch
      const int blk = min(blk0 + bpb + blk_l, lv.nblk - 1);
      const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 q = p[16 * j];
        xn[j] = (v2f){q.x, q.y};
      }
    }

    // 256-point inverse FFT of this thread's column: k = t + 16 k2, m = 16 m1 + m2
    if (!(a.pad & 2)) {
    idft16v(v);
#pragma unroll
    for (int m2 = 0; m2 < 16; ++m2) ex[t * 17 + m2] = cmulv(v[dft16_pos(m2)], tw[m2]);
    wave_sync_lds();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = ex[k1 * 17 + t];
    idft16v(v);
    }

    float* const tile = tiles + (b & 1) * kS2TileFloats;
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) {
      const v2f z = v[dft16_pos(m1)];
      const float p2 = z.x * z.x + z.y * z.y;
      tile[wbase + mstride * m1] = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
    }
    __syncthreads();

    // -- copy the finished tile out as whole lines ------------------------------
    const v4f* const tile4 = reinterpret_cast<const v4f*>(tile);
    if (!wide) {
      const int tpb = 16 << lg;                // threads per block of the batch (16 R)
      const int blk_c = tid >> (4 + lg);
      const int li = tid & (tpb - 1);
      const int blkg = blk0 + blk_c;
      if (blkg < lv.nblk) {
        const int run4 = (hop * R) >> 2;
        const int q0 = ((blk_c * 256 + halo) * R) >> 2;
        const int64_t n0 = (int64_t)blkg * hop * R;
        for (int i = li; i < run4; i += tpb) {
          int s = 0;
          if (R >= 8) {
            const int m = halo + (i >> (lg - 2));
            s = (m >> (5 - lg)) & sbits;
          }
          const v4f val = tile4[(q0 + i) ^ s];
          const int64_t n = n0 + 4 * i;
          if (a.pad & 1) { if (val[0] == 123.456f) outrow[n] = val[1]; } else
          if (vec_ok && n + 3 < a.epoch_len) {
            *reinterpret_cast<v4f*>(outrow + n) = val;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < a.epoch_len) outrow[n + e] = val[e];
          }
        }
      }
    } else if (blk0 < lv.nblk) {
      const int64_t n0 = (int64_t)blk0 * hop * R + it.rtile * 32;
      for (int idx = tid; idx < hop * 8; idx += 512) {
        const int row = idx >> 3, q = idx & 7;
        const int m = halo + row;
        const v4f val = tile4[(m * 8 + q) ^ (m & 7)];
        const int64_t n = n0 + (int64_t)row * R + 4 * q;
        if (vec_ok && n + 3 < a.epoch_len) {
          *reinterpret_cast<v4f*>(outrow + n) = val;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.epoch_len) outrow[n + e] = val[e];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_synth3: same arithmetic, different choreography.  The 16x16 transpose between
// the two DFT16 passes also re-deals the columns over the lanes, so that in the
// second pass consecutive lanes hold consecutive output samples and every wave
// store writes 256 contiguous bytes straight from registers: no staging tile.
//   pass 1: thread (column c, k1 = t)        -> U[k1][slot(c, m2)], m2 = 0..15
//   pass 2: thread id = slot(c, m2)          -> y[16 m1 + m2], m1 = 0..15
//   slot(c, m2) = blk_l*16R + m2*R + r   (R <= 32)   |   m2*32 + c   (R >= 64)
// LDS: 16 planes of 513 complex (65.7 KB) -> two workgroups per CU.
// ---------------------------------------------------------------------------
constexpr int kS3Plane = 513;
constexpr int kS3LdsBytes = 16 * kS3Plane * 8 + 256 * 8;

#ifdef GCWT_DIAG
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); acc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
template <int MODE>
__global__ void __launch_bounds__(512, 4) k_synth3(const Synth2Args a) {
#ifdef GCWT_DIAG
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);
  v2f* const twl = ex + 16 * kS3Plane;       // W256 table, 256 entries

  const Synth2Item it = a.items[blockIdx.x];
  const Synth2Level lv = a.levels[it.level];
  const int c = blockIdx.y;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo;
  const int tid = threadIdx.x;
  const int colw = tid >> 4, t = tid & 15;
  const bool wide = R > 32;
  const int bpb = wide ? 1 : (32 >> lg);
  const int blk_l = wide ? 0 : (colw >> lg);
  const int r = wide ? it.rtile * 32 + colw : (colw & (R - 1));

  if (tid < 256) {
    const float2 w = a.tw256[tid];
    twl[tid] = (v2f){w.x, w.y};
  }
  v2f hw[16];
  {
    const float2* bank = a.bank + (int64_t)it.scale * 256 + t;
    const float2* ltw = a.level_tw + lv.tw_offset;
    const float2 b0 = ltw[t * r], st = ltw[16 * r];
    v2f wcur = (v2f){b0.x, b0.y};
    const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 hk = bank[16 * j];
      hw[j] = cmulv((v2f){hk.x, hk.y}, wcur);
      wcur = cmulv(wcur, wstep);
    }
  }
  // pass-1 write position: plane t, slot(colw, m2) = wslot + m2 * sstride
  const int sstride = wide ? 32 : R;
  v2f* const exw = ex + t * kS3Plane + (wide ? colw : (blk_l << (4 + lg)) + r);
  // pass-2 identity of this thread
  const int blk_l2 = wide ? 0 : (tid >> (4 + lg));
  const int rem = wide ? tid : (tid & ((16 << lg) - 1));
  const int m2 = wide ? (tid >> 5) : (rem >> lg);
  const int r2 = wide ? it.rtile * 32 + (tid & 31) : (rem & (R - 1));
  const v2f* const exr = ex + tid;

  const float2* xb = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset + t;
  float* const outrow = a.out + ((int64_t)c * a.n_scales + it.scale) * a.n_samples + a.epoch_start;

  v2f xn[16];
  {
    const int blk = min(it.blk0 + blk_l, lv.nblk - 1);   // past-the-end columns: never stored
    const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 q = p[16 * j];
      xn[j] = (v2f){q.x, q.y};
    }
  }
  __syncthreads();                             // twl visible
  STAMP(0);

  for (int b = 0; b < it.nbatch; ++b) {
    const int blk0 = it.blk0 + b * bpb;
    v2f v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmulv(xn[j], hw[j]);
    STAMP(1);
    if (b + 1 < it.nbatch && !(a.pad & 4)) {   // next batch's spectra, ahead of this batch's stores
      const int blk = min(blk0 + bpb + blk_l, lv.nblk - 1);
      const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 q = p[16 * j];
        xn[j] = (v2f){q.x, q.y};
      }
    }
    if (!(a.pad & 2)) idft16v(v);
#pragma unroll
    for (int j = 0; j < 16; ++j) if (!(a.pad & 16)) exw[j * sstride] = cmulv(v[dft16_pos(j)], twl[(t * j) & 255]);
    STAMP(2);
    if (!(a.pad & 8)) __syncthreads();
    STAMP(3);
    if (!(a.pad & 16))
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = exr[k1 * kS3Plane];
    STAMP(4);
    if (!(a.pad & 8)) __syncthreads();
    STAMP(5);
    if (!(a.pad & 2)) idft16v(v);

    const int blkg = blk0 + blk_l2;
    // sample of (m1 = 0): n = blkg*hop*R + (m2 - halo)*R + r2 ; + 16 R per m1
    const int64_t n0 = ((int64_t)blkg * hop + (m2 - halo)) * R + r2;
    const int64_t n_lo = (int64_t)blkg * hop * R;
    const int64_t n_hi = min(n_lo + (int64_t)hop * R, a.epoch_len);
    if (blkg < lv.nblk) {
#pragma unroll
      for (int m1 = 0; m1 < 16; ++m1) {
        const int64_t n = n0 + (int64_t)(m1 * 16) * R;
        const v2f z = v[dft16_pos(m1)];
        const float p2 = z.x * z.x + z.y * z.y;
        if (n >= n_lo && n < n_hi && (!(a.pad & 1) || p2 == 123.456f))
          outrow[n] = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
      }
    }
    STAMP(6);
  }
#ifdef GCWT_DIAG
  if ((tid & 63) == 0 && a.diag) {
    for (int i = 0; i < 8; ++i) atomicAdd(a.diag + i, acc[i]);
    atomicAdd(a.diag + 8, 1ull);
  }
#endif
}

hipError_t launch_synth3(int mode, const Synth2Args& a, int n_items, int n_channels, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synth3<GCWT_OUT_AMPLITUDE_F32>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kS3LdsBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth3<GCWT_OUT_POWER_F32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, kS3LdsBytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(512);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth3<GCWT_OUT_AMPLITUDE_F32>), grid, block, kS3LdsBytes, st, a);
  else
    hipLaunchKernelGGL((k_synth3<GCWT_OUT_POWER_F32>), grid, block, kS3LdsBytes, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_synth4<MODE, NCOL>: k_synth3's choreography with NCOL columns per batch
// (16 * NCOL threads), scalar-uniform row masks instead of per-store 64-bit
// predicates, and 32-bit offsets from a wave-uniform row pointer.
// NCOL = 16: LDS 34.9 KB -> four workgroups per CU.
// ---------------------------------------------------------------------------
template <int MODE, int NCOL>
__global__ void __launch_bounds__(16 * NCOL, (NCOL == 16 ? 3 : 2)) k_synth4(const Synth2Args a) {
  constexpr int kThreads = 16 * NCOL;
  constexpr int kPlane = kThreads + 1;
  constexpr int kLgN = NCOL == 16 ? 4 : 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);
  v2f* const twl = ex + 16 * kPlane;         // W256 table, 256 entries

  const Synth2Item it = a.items[blockIdx.x];
  const Synth2Level lv = a.levels[it.level];
  const int c = blockIdx.y;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo;
  const int tid = threadIdx.x;
  const int colw = tid >> 4, t = tid & 15;
  const bool wide = R > NCOL;                // one block per batch, NCOL of its R phases
  const int bpb = wide ? 1 : (NCOL >> lg);
  const int blk_l = wide ? 0 : (colw >> lg);
  const int r = wide ? it.rtile * NCOL + colw : (colw & (R - 1));

  for (int i = tid; i < 256; i += kThreads) {
    const float2 w = a.tw256[i];
    twl[i] = (v2f){w.x, w.y};
  }
  v2f hw[16];
  {
    const float2* bank = a.bank + (int64_t)it.scale * 256 + t;
    const float2* ltw = a.level_tw + lv.tw_offset;
    const float2 b0 = ltw[t * r], st = ltw[16 * r];
    v2f wcur = (v2f){b0.x, b0.y};
    const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 hk = bank[16 * j];
      hw[j] = cmulv((v2f){hk.x, hk.y}, wcur);
      wcur = cmulv(wcur, wstep);
    }
  }
  const int sstride = wide ? NCOL : R;
  v2f* const exw = ex + t * kPlane + (wide ? colw : (blk_l << (4 + lg)) + r);
  const int blk_l2 = wide ? 0 : (tid >> (4 + lg));
  const int rem = wide ? tid : (tid & ((16 << lg) - 1));
  const int m2 = wide ? (tid >> kLgN) : (rem >> lg);
  const int r2 = wide ? it.rtile * NCOL + (tid & (NCOL - 1)) : (rem & (R - 1));
  const v2f* const exr = ex + tid;
  // offset of this thread's (m1 = 0) sample from the first sample of the batch's first block
  const int off0 = (blk_l2 * hop + m2 - halo) * R + r2;
  const int m1step = 16 * R;

  const float2* xb = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset + t;
  float* const outrow = a.out + ((int64_t)c * a.n_scales + it.scale) * a.n_samples + a.epoch_start;

  v2f xn[16];
  {
    const int blk = min(it.blk0 + blk_l, lv.nblk - 1);   // past-the-end columns: never stored
    const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 q = p[16 * j];
      xn[j] = (v2f){q.x, q.y};
    }
  }
  __syncthreads();                             // twl visible

  for (int b = 0; b < it.nbatch; ++b) {
    const int blk0 = it.blk0 + b * bpb;
    v2f v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmulv(xn[j], hw[j]);
    if (b + 1 < it.nbatch) {                   // next batch's spectra, ahead of this batch's stores
      const int blk = min(blk0 + bpb + blk_l, lv.nblk - 1);
      const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 q = p[16 * j];
        xn[j] = (v2f){q.x, q.y};
      }
    }
    idft16v(v);
#pragma unroll
    for (int j = 0; j < 16; ++j) exw[j * sstride] = cmulv(v[dft16_pos(j)], twl[(t * j) & 255]);
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = exr[k1 * kPlane];
    __syncthreads();
    idft16v(v);

    const int64_t n_b = (int64_t)blk0 * hop * R;           // first sample of the batch (uniform)
    float* const dst = outrow + n_b;
    const int span = bpb * hop * R;                        // samples the batch covers
    const bool inside = blk0 + bpb <= lv.nblk && n_b + span <= a.epoch_len;
    // samples this thread may write: its block exists and the sample is inside the epoch
    const int lim = inside ? 0x7fffffff
                           : (blk0 + blk_l2 < lv.nblk ? (int)min<int64_t>(a.epoch_len - n_b, 0x7fffffff) : 0);
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) {
      // rows 16 m1 .. 16 m1 + 15 of the block: all kept, none kept, or split (uniform tests)
      if (16 * m1 + 15 < halo || 16 * m1 >= halo + hop) continue;
      const v2f z = v[dft16_pos(m1)];
      const float p2 = z.x * z.x + z.y * z.y;
      const float val = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
      const int off = off0 + m1 * m1step;
      const int m = 16 * m1 + m2;
      const bool rows_ok = (16 * m1 >= halo && 16 * m1 + 15 < halo + hop) || (m >= halo && m < halo + hop);
      if (rows_ok && off < lim) dst[off] = val;
    }
  }
}

template <int NCOL>
static hipError_t launch_synth4_n(int mode, const Synth2Args& a, int n_items, int n_channels,
                                  hipStream_t st) {
  constexpr int lds = 16 * (16 * NCOL + 1) * 8 + 256 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synth4<GCWT_OUT_AMPLITUDE_F32, NCOL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth4<GCWT_OUT_POWER_F32, NCOL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(16 * NCOL);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth4<GCWT_OUT_AMPLITUDE_F32, NCOL>), grid, block, lds, st, a);
  else
    hipLaunchKernelGGL((k_synth4<GCWT_OUT_POWER_F32, NCOL>), grid, block, lds, st, a);
  return hipGetLastError();
}

hipError_t launch_synth4(int mode, int ncol, const Synth2Args& a, int n_items, int n_channels,
                         hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  return ncol == 16 ? launch_synth4_n<16>(mode, a, n_items, n_channels, st)
                    : launch_synth4_n<32>(mode, a, n_items, n_channels, st);
}

// ---------------------------------------------------------------------------
// k_synth5<MODE>: production kernel.  k_synth3's choreography (32 columns, pass 2
// re-dealt so that lanes hold consecutive samples) specialised for block layouts
// with 16 <= halo <= 32: of the 16 output rows a thread holds (m = 16 m1 + m2),
// rows 0 and 15 are always halo, rows 2..13 are always kept, rows 1 and 14 are
// kept by the lanes with m2 >= halo - 16 / m2 < 32 - halo.  Stores use a
// wave-uniform row pointer plus a 32-bit lane offset; batches that touch the end
// of the epoch or of the block list take a checked path.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(512, 4) k_synth5(const Synth2Args a) {
  constexpr int kPlane = 513;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);
  v2f* const twl = ex + 16 * kPlane;

  const Synth2Item it = a.items[blockIdx.x];
  const Synth2Level lv = a.levels[it.level];
  const int c = blockIdx.y;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo;
  const int tid = threadIdx.x;
  const int colw = tid >> 4, t = tid & 15;
  const bool wide = R > 32;
  const int bpb = wide ? 1 : (32 >> lg);
  const int blk_l = wide ? 0 : (colw >> lg);
  const int r = wide ? it.rtile * 32 + colw : (colw & (R - 1));

  if (tid < 256) {
    const float2 w = a.tw256[tid];
    twl[tid] = (v2f){w.x, w.y};
  }
  v2f hw[16];
  {
    const float2* bank = a.bank + (int64_t)it.scale * 256 + t;
    const float2* ltw = a.level_tw + lv.tw_offset;
    const float2 b0 = ltw[t * r], st = ltw[16 * r];
    v2f wcur = (v2f){b0.x, b0.y};
    const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 hk = bank[16 * j];
      hw[j] = cmulv((v2f){hk.x, hk.y}, wcur);
      wcur = cmulv(wcur, wstep);
    }
  }
  const int sstride = wide ? 32 : R;
  v2f* const exw = ex + t * kPlane + (wide ? colw : (blk_l << (4 + lg)) + r);
  const int blk_l2 = wide ? 0 : (tid >> (4 + lg));
  const int rem = wide ? tid : (tid & ((16 << lg) - 1));
  const int m2 = wide ? (tid >> 5) : (rem >> lg);
  const int r2 = wide ? it.rtile * 32 + (tid & 31) : (rem & (R - 1));
  const v2f* const exr = ex + tid;
  const int off0 = (blk_l2 * hop + m2 - halo) * R + r2;   // sample offset of row m1 = 0
  const int m1step = 16 * R;
  const bool keep1 = m2 >= halo - 16, keep14 = m2 < 32 - halo;

  const float2* xb = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset + t;
  float* const outrow = a.out + ((int64_t)c * a.n_scales + it.scale) * a.n_samples + a.epoch_start;

  v2f xn[16];
  {
    const int blk = min(it.blk0 + blk_l, lv.nblk - 1);   // past-the-end columns: never stored
    const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 q = p[16 * j];
      xn[j] = (v2f){q.x, q.y};
    }
  }
  __syncthreads();

  for (int b = 0; b < it.nbatch; ++b) {
    const int blk0 = it.blk0 + b * bpb;
    v2f v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmulv(xn[j], hw[j]);
    if (b + 1 < it.nbatch) {
      const int blk = min(blk0 + bpb + blk_l, lv.nblk - 1);
      const float2* p = xb + (int64_t)blk * 256;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 q = p[16 * j];
        xn[j] = (v2f){q.x, q.y};
      }
    }
    idft16v(v);
#pragma unroll
    for (int j = 0; j < 16; ++j) exw[j * sstride] = cmulv(v[dft16_pos(j)], twl[(t * j) & 255]);
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = exr[k1 * kPlane];
    __syncthreads();
    idft16v(v);

    const int64_t n_b = (int64_t)blk0 * hop * R;
    float* const dst = outrow + n_b;
    const int span = bpb * hop * R;
    const bool inside = blk0 + bpb <= lv.nblk && n_b + span <= a.epoch_len;   // wave-uniform
    auto mag = [&](int m1) {
      const v2f z = v[dft16_pos(m1)];
      const float p2 = z.x * z.x + z.y * z.y;
      return MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
    };
    if (inside) {
      if (keep1) dst[off0 + m1step] = mag(1);
#pragma unroll
      for (int m1 = 2; m1 < 14; ++m1) dst[off0 + m1 * m1step] = mag(m1);
      if (keep14) dst[off0 + 14 * m1step] = mag(14);
    } else {
      const int lim = blk0 + blk_l2 < lv.nblk
                          ? (int)min<int64_t>(a.epoch_len - n_b, (int64_t)0x7fffffff) : 0;
#pragma unroll
      for (int m1 = 1; m1 < 15; ++m1) {
        const bool keep = m1 == 1 ? keep1 : (m1 == 14 ? keep14 : true);
        const int off = off0 + m1 * m1step;
        if (keep && off < lim) dst[off] = mag(m1);
      }
    }
  }
}

hipError_t launch_synth5(int mode, const Synth2Args& a, int n_items, int n_channels, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  constexpr int lds = 16 * 513 * 8 + 256 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synth5<GCWT_OUT_AMPLITUDE_F32>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth5<GCWT_OUT_POWER_F32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(512);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth5<GCWT_OUT_AMPLITUDE_F32>), grid, block, lds, st, a);
  else
    hipLaunchKernelGGL((k_synth5<GCWT_OUT_POWER_F32>), grid, block, lds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_synth6<MODE, STAGE>: k_synth5 with the register budget brought under 128
// VGPRs (two 512-thread workgroups per CU, four waves per SIMD).
// STAGE = true  (R >= 8): the batch's block spectra (<= 4 blocks, 8 KB) are
//   fetched with one or two coalesced loads per thread a batch ahead and parked in
//   LDS; each thread then reads its 16 inputs as LDS broadcasts.  This also takes
//   the 8..32-fold redundant spectrum reads off the vector L1.
// STAGE = false (R <= 4): every thread reads its own 16 inputs from global memory;
//   the loads for the next batch are issued after the second DFT, ahead of the
//   stores, so they do not queue behind them (vmcnt retires in order).
// ---------------------------------------------------------------------------
template <int MODE, bool STAGE>
__global__ void __launch_bounds__(512, 4) k_synth6(const Synth2Args a) {
  constexpr int kPlane = 513;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);
  v2f* const twl = ex + 16 * kPlane;
  v2f* const stage = twl + 256;              // STAGE: up to 4 blocks x 256

  const Synth2Item it = a.items[blockIdx.x];
  const Synth2Level lv = a.levels[it.level];
  const int c = blockIdx.y;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo;
  const int tid = threadIdx.x;
  const int colw = tid >> 4, t = tid & 15;
  const bool wide = R > 32;
  const int bpb = wide ? 1 : (32 >> lg);
  const int blk_l = wide ? 0 : (colw >> lg);
  const int r = wide ? it.rtile * 32 + colw : (colw & (R - 1));

  if (tid < 256) {
    const float2 w = a.tw256[tid];
    twl[tid] = (v2f){w.x, w.y};
  }
  v2f hw[16];
  {
    const float2* bank = a.bank + (int64_t)it.scale * 256 + t;
    const float2* ltw = a.level_tw + lv.tw_offset;
    const float2 b0 = ltw[t * r], st = ltw[16 * r];
    v2f wcur = (v2f){b0.x, b0.y};
    const v2f wstep = (v2f){st.x, st.y};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 hk = bank[16 * j];
      hw[j] = cmulv((v2f){hk.x, hk.y}, wcur);
      wcur = cmulv(wcur, wstep);
    }
  }
  const int sstride = wide ? 32 : R;
  v2f* const exw = ex + t * kPlane + (wide ? colw : (blk_l << (4 + lg)) + r);
  const int blk_l2 = wide ? 0 : (tid >> (4 + lg));
  const int rem = wide ? tid : (tid & ((16 << lg) - 1));
  const int m2 = wide ? (tid >> 5) : (rem >> lg);
  const int r2 = wide ? it.rtile * 32 + (tid & 31) : (rem & (R - 1));
  const v2f* const exr = ex + tid;
  const int off0 = (blk_l2 * hop + m2 - halo) * R + r2;   // sample offset of row m1 = 0
  const int m1step = 16 * R;
  const bool keep1 = m2 >= halo - 16, keep14 = m2 < 32 - halo;

  const float2* const xbl = a.xb + (int64_t)c * a.xb_cstride + lv.xb_offset;  // level spectra
  const int64_t xb_last = (int64_t)lv.nblk * 256 - 1;
  float* const outrow = a.out + ((int64_t)c * a.n_scales + it.scale) * a.n_samples + a.epoch_start;
  const v2f* const st_rd = stage + blk_l * 256 + t;
  const int n_stage = bpb * 256;             // complex values staged per batch (256 .. 1024)

  v2f xn[16];
  if (STAGE) {
    for (int i = tid; i < n_stage; i += 512) {
      const float2 q = xbl[min((int64_t)it.blk0 * 256 + i, xb_last)];
      stage[i] = (v2f){q.x, q.y};
    }
  } else {
    const int blk = min(it.blk0 + blk_l, lv.nblk - 1);   // past-the-end columns: never stored
    const float2* p = xbl + (int64_t)blk * 256 + t;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 q = p[16 * j];
      xn[j] = (v2f){q.x, q.y};
    }
  }
  __syncthreads();

  for (int b = 0; b < it.nbatch; ++b) {
    const int blk0 = it.blk0 + b * bpb;
    const bool more = b + 1 < it.nbatch;
    v2f v[16];
    float2 g0 = make_float2(0.f, 0.f), g1 = g0;
    if (STAGE) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = cmulv(st_rd[16 * j], hw[j]);
      if (more) {                              // next batch's spectra: in flight during this batch
        const int64_t base = (int64_t)(blk0 + bpb) * 256;
        if (tid < n_stage) g0 = xbl[min(base + tid, xb_last)];
        if (tid + 512 < n_stage) g1 = xbl[min(base + tid + 512, xb_last)];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = cmulv(xn[j], hw[j]);
    }
    idft16v(v);
#pragma unroll
    for (int j = 0; j < 16; ++j) exw[j * sstride] = cmulv(v[dft16_pos(j)], twl[(t * j) & 255]);
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = exr[k1 * kPlane];
    if (STAGE && more) {                       // everyone has read this batch's stage by now
      if (tid < n_stage) stage[tid] = (v2f){g0.x, g0.y};
      if (tid + 512 < n_stage) stage[tid + 512] = (v2f){g1.x, g1.y};
    }
    __syncthreads();
    idft16v(v);
    if (!STAGE && more) {                      // issue ahead of the stores below
      const int blk = min(blk0 + bpb + blk_l, lv.nblk - 1);
      const float2* p = xbl + (int64_t)blk * 256 + t;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 q = p[16 * j];
        xn[j] = (v2f){q.x, q.y};
      }
    }

    const int64_t n_b = (int64_t)blk0 * hop * R;
    float* const dst = outrow + n_b;
    const int span = bpb * hop * R;
    const bool inside = blk0 + bpb <= lv.nblk && n_b + span <= a.epoch_len;   // wave-uniform
    const int lim = inside ? 0x7fffffff
                           : (blk0 + blk_l2 < lv.nblk
                                  ? (int)min<int64_t>(a.epoch_len - n_b, (int64_t)0x7fffffff) : 0);
#pragma unroll
    for (int m1 = 1; m1 < 15; ++m1) {
      const v2f z = v[dft16_pos(m1)];
      const float p2 = z.x * z.x + z.y * z.y;
      const float val = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
      const bool keep = m1 == 1 ? keep1 : (m1 == 14 ? keep14 : true);
      const int off = off0 + m1 * m1step;
      if (keep && off < lim) dst[off] = val;
    }
  }
}

hipError_t launch_synth6(int mode, const Synth2Args& a, int n_items, int n_channels, bool staged,
                         hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  constexpr int lds = 16 * 513 * 8 + 256 * 8 + 1024 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[4] = {(const void*)k_synth6<GCWT_OUT_AMPLITUDE_F32, true>,
                          (const void*)k_synth6<GCWT_OUT_AMPLITUDE_F32, false>,
                          (const void*)k_synth6<GCWT_OUT_POWER_F32, true>,
                          (const void*)k_synth6<GCWT_OUT_POWER_F32, false>};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
    }
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(512);
  if (mode == GCWT_OUT_AMPLITUDE_F32) {
    if (staged) hipLaunchKernelGGL((k_synth6<GCWT_OUT_AMPLITUDE_F32, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_synth6<GCWT_OUT_AMPLITUDE_F32, false>), grid, block, lds, st, a);
  } else {
    if (staged) hipLaunchKernelGGL((k_synth6<GCWT_OUT_POWER_F32, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_synth6<GCWT_OUT_POWER_F32, false>), grid, block, lds, st, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_synth7<MODE>: the production synthesis kernel.
//
// One workgroup owns a fixed set of 32 columns -- 32/R consecutive blocks x all R
// phases (R <= 32), or one block x 32 of its phases (R >= 64) -- and walks over
// the SCALES of the level.  What never changes for a thread,
//     P[k] = XB_blk[k] * W^{k r},   k = t + 16 j,
// is built once and kept in 32 VGPRs; what changes per batch, the scale's filter
// H_s[k] (2 KB), is fetched one batch ahead by the first 256 threads and parked in
// LDS, from where every column reads it as a broadcast.  Per batch and thread:
//   v = P * H_s          16 complex multiplies
//   DFT16, W256 twiddle (table in LDS), transpose + re-deal through LDS
//   DFT16, |.|, 14 stores of 4 B per lane: 256 contiguous bytes per wave store
// With 16 <= halo <= 32 rows 0 and 15 of a thread's 16 outputs are always halo,
// rows 2..13 are always kept and rows 1 / 14 are kept lane-wise.
// LDS: 16 x 513 complex + 2 x 256 complex = 69.8 KB -> two workgroups per CU.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(512, 4) k_synth7(const Synth7Args a) {
  constexpr int kPlane = 513;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* const ex = reinterpret_cast<v2f*>(smem);
  v2f* const twl = ex + 16 * kPlane;
  v2f* const stage = twl + 256;

  const Synth7Item it = a.items[blockIdx.x];
  const Synth7Level lv = a.levels[it.level];
  const int c = blockIdx.y;
  const int R = lv.decimation, lg = lv.log2r, hop = lv.hop, halo = lv.halo;
  const int tid = threadIdx.x;
  const int colw = tid >> 4, t = tid & 15;
  const bool wide = R > 32;
  const int bpb = wide ? 1 : (32 >> lg);
  const int blk_l = wide ? 0 : (colw >> lg);
  const int r = wide ? it.rtile * 32 + colw : (colw & (R - 1));
  const int* const scales = a.scale_list + lv.scale_offset;

  if (tid < 256) {
    const float2 w = a.tw256[tid];
    twl[tid] = (v2f){w.x, w.y};
    const float2 h = a.bank[(int64_t)scales[0] * 256 + tid];
    stage[tid] = (v2f){h.x, h.y};
  }
  v2f pw[16];
  {
    const int blk = min(it.blk0 + blk_l, lv.nblk - 1);   // past-the-end columns: never stored
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
  const int sstride = wide ? 32 : R;
  v2f* const exw = ex + t * kPlane + (wide ? colw : (blk_l << (4 + lg)) + r);
  const int blk_l2 = wide ? 0 : (tid >> (4 + lg));
  const int rem = wide ? tid : (tid & ((16 << lg) - 1));
  const int m2 = wide ? (tid >> 5) : (rem >> lg);
  const int r2 = wide ? it.rtile * 32 + (tid & 31) : (rem & (R - 1));
  const v2f* const exr = ex + tid;
  const int off0 = (blk_l2 * hop + m2 - halo) * R + r2;   // sample offset of row m1 = 0
  const int m1step = 16 * R;
  const bool keep1 = m2 >= halo - 16, keep14 = m2 < 32 - halo;

  const int64_t n_b = (int64_t)it.blk0 * hop * R;          // first sample of the block group
  const int span = bpb * hop * R;
  const bool inside = it.blk0 + bpb <= lv.nblk && n_b + span <= a.epoch_len;   // wave-uniform
  const int lim = inside ? 0x7fffffff
                         : (it.blk0 + blk_l2 < lv.nblk
                                ? (int)min<int64_t>(a.epoch_len - n_b, (int64_t)0x7fffffff) : 0);
  float* const out0 = a.out + (int64_t)c * a.n_scales * a.n_samples + a.epoch_start + n_b;
  const v2f* const st_rd = stage + t;
  __syncthreads();

  for (int b = 0; b < lv.n_scales; ++b) {
    const bool more = b + 1 < lv.n_scales;
    v2f* const st_cur = stage;                 // single buffer: refilled between the barriers
    v2f v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmulv(st_rd[16 * j], pw[j]);
    float2 g = make_float2(0.f, 0.f);
    if (more && tid < 256) g = a.bank[(int64_t)scales[b + 1] * 256 + tid];
    idft16v(v);
#pragma unroll
    for (int j = 0; j < 16; ++j) exw[j * sstride] = cmulv(v[dft16_pos(j)], twl[(t * j) & 255]);
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = exr[k1 * kPlane];
    if (more && tid < 256) st_cur[tid] = (v2f){g.x, g.y};   // every thread is past this batch's reads
    __syncthreads();
    idft16v(v);

    float* const dst = out0 + (int64_t)scales[b] * a.n_samples;
#pragma unroll
    for (int m1 = 1; m1 < 15; ++m1) {
      const v2f z = v[dft16_pos(m1)];
      const float p2 = z.x * z.x + z.y * z.y;
      const float val = MODE == GCWT_OUT_AMPLITUDE_F32 ? __builtin_amdgcn_sqrtf(p2) : p2;
      const bool keep = m1 == 1 ? keep1 : (m1 == 14 ? keep14 : true);
      const int off = off0 + m1 * m1step;
      if (keep && off < lim) dst[off] = val;
    }
  }
}

hipError_t launch_synth7(int mode, const Synth7Args& a, int n_items, int n_channels, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  constexpr int lds = 16 * 513 * 8 + 2 * 256 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synth7<GCWT_OUT_AMPLITUDE_F32>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth7<GCWT_OUT_POWER_F32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(512);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth7<GCWT_OUT_AMPLITUDE_F32>), grid, block, lds, st, a);
  else
    hipLaunchKernelGGL((k_synth7<GCWT_OUT_POWER_F32>), grid, block, lds, st, a);
  return hipGetLastError();
}

hipError_t launch_synth2(int mode, const Synth2Args& a, int n_items, int n_channels, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_synth2<GCWT_OUT_AMPLITUDE_F32>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kS2LdsBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_synth2<GCWT_OUT_POWER_F32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, kS2LdsBytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(n_items, n_channels), block(512);
  if (mode == GCWT_OUT_AMPLITUDE_F32)
    hipLaunchKernelGGL((k_synth2<GCWT_OUT_AMPLITUDE_F32>), grid, block, kS2LdsBytes, st, a);
  else
    hipLaunchKernelGGL((k_synth2<GCWT_OUT_POWER_F32>), grid, block, kS2LdsBytes, st, a);
  return hipGetLastError();
}

}  // namespace gcwt
