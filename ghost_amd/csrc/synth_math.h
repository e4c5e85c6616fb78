// synth_math.h -- packed complex arithmetic and the register DFT16 shared by the
// synthesis kernels (synth.hip, synth8.hip).
#pragma once
#include <hip/hip_runtime.h>

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

// First-layer butterfly of idft16v whose inputs c and / or d are known to be zero at compile
// time: 4 packed adds with both zero, 6 with d zero, 8 with neither.
template <bool CZ, bool DZ>
__device__ __forceinline__ void bfly4_in(v2f& a, v2f& b, v2f& c, v2f& d) {
  static_assert(DZ || !CZ, "the window is cut from the top: c zero implies d zero");
  v2f s0, s1, s2, u;
  if constexpr (CZ) { s0 = a; s1 = a; } else { s0 = a + c; s1 = a - c; }
  if constexpr (DZ) { s2 = b; u = b; } else { s2 = b + d; u = b - d; }
  a = s0 + s2;
  c = s0 - s2;
  b = add_ib(s1, u);
  d = sub_ib(s1, u);
}

// v[j] = p[j] * g[j] for the inputs j < JHI (the others are structurally zero: the scale's
// gain is negligible on bins 16 j .. 16 j + 15 for j >= JHI), then the first radix-4 layer of
// idft16v without the work the zeros would cost: 3 packed instructions per zero input.
// g: the lane's 16 gains as four float4.
template <int JHI>
__device__ __forceinline__ void gain_first_layer(v2f v[16], const v2f p[16], const float4* g) {
  static_assert(JHI >= 9 && JHI <= 16, "inputs 0..8 are always computed");
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (4 * q >= JHI) continue;
    const float4 gq = g[q];
    v[4 * q] = p[4 * q] * gq.x;
    if (4 * q + 1 < JHI) v[4 * q + 1] = p[4 * q + 1] * gq.y;
    if (4 * q + 2 < JHI) v[4 * q + 2] = p[4 * q + 2] * gq.z;
    if (4 * q + 3 < JHI) v[4 * q + 3] = p[4 * q + 3] * gq.w;
  }
  bfly4_in<(8 >= JHI), (12 >= JHI)>(v[0], v[4], v[8], v[12]);
  bfly4_in<(9 >= JHI), (13 >= JHI)>(v[1], v[5], v[9], v[13]);
  bfly4_in<(10 >= JHI), (14 >= JHI)>(v[2], v[6], v[10], v[14]);
  bfly4_in<(11 >= JHI), (15 >= JHI)>(v[3], v[7], v[11], v[15]);
}

// the rest of idft16v after its first layer: twiddles W16^(n1 k2), second radix-4 layer
__device__ __forceinline__ void idft16v_tail(v2f v[16]) {
  const v2f w1 = {0.92387953251128674f, 0.38268343236508977f};   // W16^1
  const v2f w3 = {0.38268343236508977f, 0.92387953251128674f};   // W16^3
  const v2f w9 = {-0.92387953251128674f, -0.38268343236508977f}; // W16^9
  const float h = 0.70710678118654752f;
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

__device__ __forceinline__ void idft16v(v2f v[16]) {
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) bfly4<false>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
  idft16v_tail(v);
}

}  // namespace gcwt
