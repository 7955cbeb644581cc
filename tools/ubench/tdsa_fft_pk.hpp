// tdsa_fft_pk.hpp - packed-fp32 (VOP3P) radix butterflies for the gfx950 frame kernel.
//
// A complex value lives in an even-aligned VGPR pair and all complex arithmetic is issued as
// v_pk_{add,mul,fma}_f32: measured on MI355X (tools/ubench/valu_dep.hip) a packed op costs 1.8 ns per
// wave-instruction per SIMD whatever the instruction-level parallelism, i.e. 0.9 ns per real
// operation, where the scalar v_add/v_mul/v_fma stream of a butterfly network sustains 1.2-1.7 ns.
// The per-half modifiers of VOP3P (op_sel / op_sel_hi / neg_lo / neg_hi) make the multiplications by
// -i and the (re, im) swizzle of a complex product free, so
//     a + b, a - b                       1 instruction each
//     a -/+ i b                          1 instruction each (no separate rotation)
//     a * w  (w in a register pair)      2 instructions
//     a * W_R^K (compile-time constant)  2 instructions, the (cos, sin) pair sits in SGPRs
// The layer structure is the radix-2 DIF network of tdsa_fft.hpp with consecutive layers fused into
// radix-4 butterflies; results land in the same (bit-reversed) registers.
// Replaces np.fft.fft / scipy.fft.fft (datasources/hackrf_samples.py:370, rtl_samples.py:170).
#pragma once
#include "../../topdogspectrumanalyser_amd/csrc/tdsa_fft.hpp"

namespace tdsa {

typedef float p32 __attribute__((ext_vector_type(2)));   // (re, im) in one 64-bit VGPR pair

__device__ __forceinline__ p32 to_p(c32 a) { return p32{a.x, a.y}; }
__device__ __forceinline__ c32 to_c(p32 a) { return c32{a.x, a.y}; }

__device__ __forceinline__ p32 padd(p32 a, p32 b) { return a + b; }     // v_pk_add_f32
__device__ __forceinline__ p32 psub(p32 a, p32 b) { return a - b; }     // v_pk_add_f32 neg_lo/neg_hi
// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ p32 padd_mi(p32 a, p32 b) {
  p32 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ p32 padd_pi(p32 a, p32 b) {
  p32 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a * w, both in VGPR pairs
__device__ __forceinline__ p32 pmul(p32 a, p32 w) {
  p32 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                     // a * w.x
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"             // (-a.y w.y, a.x w.y) + t
      : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// a * (c - i s) with the constant pair (c, s) in SGPRs: (a.x c + a.y s, a.y c - a.x s)
__device__ __forceinline__ p32 pmul_cs(p32 a, p32 cs) {
  p32 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(cs));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(cs), "v"(t));
  return r;
}
// -i a, +i a, -a as stand-alone operations (only where they cannot be folded into an add)
__device__ __forceinline__ p32 pmul_mi(p32 a) { return p32{a.y, -a.x}; }
__device__ __forceinline__ p32 pmul_pi(p32 a) { return p32{-a.y, a.x}; }

// a * W_R^K, K and R compile-time
template <int K, int R>
__device__ __forceinline__ p32 pmul_w(p32 a) {
  constexpr int k = ((K % R) + R) % R;
  if constexpr (k == 0) {
    return a;
  } else if constexpr (4 * k == R) {
    return pmul_mi(a);
  } else if constexpr (2 * k == R) {
    return -a;
  } else if constexpr (4 * k == 3 * R) {
    return pmul_pi(a);
  } else {
    constexpr float c = float(cx_cos(cx_angle(k, R)));
    constexpr float s = float(cx_sin(cx_angle(k, R)));
    return pmul_cs(a, p32{c, s});
  }
}

// In-place radix-R DIF on v[BASE .. BASE+R): afterwards X[k] = v[BASE + bitrev(k, log2 R)], exactly the
// register layout of dif<R, BASE, TOT>.  Two radix-2 layers at a time:
//   a, b, c, d = v[i], v[i+q], v[i+2q], v[i+3q]   (q = R/4)
//   v[i]    = (a + c) + (b + d)
//   v[i+q]  = ((a + c) - (b + d)) W_R^(2i)
//   v[i+2q] = ((a - c) - i (b - d)) W_R^i
//   v[i+3q] = ((a - c) + i (b - d)) W_R^(3i)
template <int R, int BASE, int TOT>
__device__ __forceinline__ void difp(p32 (&v)[TOT]) {
  if constexpr (R == 2) {
    const p32 a = v[BASE], b = v[BASE + 1];
    v[BASE] = padd(a, b);
    v[BASE + 1] = psub(a, b);
  } else if constexpr (R >= 4) {
    constexpr int q = R / 4;
    static_for<0, q>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const p32 a = v[BASE + i], b = v[BASE + i + q], c = v[BASE + i + 2 * q], d = v[BASE + i + 3 * q];
      const p32 s0 = padd(a, c), d0 = psub(a, c), s1 = padd(b, d), d1 = psub(b, d);
      v[BASE + i] = padd(s0, s1);
      v[BASE + i + q] = pmul_w<2 * i, R>(psub(s0, s1));
      v[BASE + i + 2 * q] = pmul_w<i, R>(padd_mi(d0, d1));
      v[BASE + i + 3 * q] = pmul_w<3 * i, R>(padd_pi(d0, d1));
    });
    static_for<0, 4>([&](auto jc) { difp<q, BASE + decltype(jc)::value * q, TOT>(v); });
  }
}

}  // namespace tdsa
