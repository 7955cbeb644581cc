// tdsa_fft.hpp - in-register radix butterflies for the gfx950 spectrum kernels.
//
// Everything here works on a per-thread array of complex values that the compiler keeps in VGPRs
// (all indices are compile-time constants after full unrolling).  A radix-R DFT is done as log2(R)
// decimation-in-frequency radix-2 layers with the trivial twiddles (1, -i, (+-1-i)/sqrt2) folded at
// compile time, leaving the result in bit-reversed register order; callers undo the permutation
// for free by indexing with bitrev().  Replaces np.fft.fft / scipy.fft.fft
// (datasources/hackrf_samples.py:370, datasources/rtl_samples.py:170 of the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace tdsa {

using c32 = float2;

__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return c32{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return c32{a.x - b.x, a.y - b.y}; }
// a * b
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
  return c32{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// ---- compile-time loop -------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- compile-time trig (double Taylor series; only used to make float constants) ----------------
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double cx_sin(double x) {  // |x| <= pi
  double x2 = x * x, term = x, sum = x;
  for (int i = 1; i <= 16; ++i) {
    term *= -x2 / double((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
constexpr double cx_cos(double x) {  // |x| <= pi
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int i = 1; i <= 16; ++i) {
    term *= -x2 / double((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
// angle 2*pi*k/n reduced to (-pi, pi]
constexpr double cx_angle(int k, int n) {
  int kk = ((k % n) + n) % n;
  if (2 * kk > n) kk -= n;
  return 2.0 * kPi * double(kk) / double(n);
}

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }
constexpr int bitrev(int x, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

// v * exp(-2*pi*i*K/R), K and R compile-time
template <int K, int R>
__device__ __forceinline__ c32 mul_w(c32 v) {
  constexpr int k = ((K % R) + R) % R;
  constexpr float s8 = 0.70710678118654752440f;
  if constexpr (k == 0) {
    return v;
  } else if constexpr (4 * k == R) {  // -i
    return c32{v.y, -v.x};
  } else if constexpr (2 * k == R) {  // -1
    return c32{-v.x, -v.y};
  } else if constexpr (4 * k == 3 * R) {  // +i
    return c32{-v.y, v.x};
  } else if constexpr (8 * k == R) {  // (1 - i)/sqrt2
    return c32{(v.x + v.y) * s8, (v.y - v.x) * s8};
  } else if constexpr (8 * k == 3 * R) {  // (-1 - i)/sqrt2
    return c32{(v.y - v.x) * s8, -(v.x + v.y) * s8};
  } else if constexpr (8 * k == 5 * R) {  // (-1 + i)/sqrt2
    return c32{-(v.x + v.y) * s8, (v.x - v.y) * s8};
  } else if constexpr (8 * k == 7 * R) {  // (1 + i)/sqrt2
    return c32{(v.x - v.y) * s8, (v.x + v.y) * s8};
  } else {
    constexpr float c = float(cx_cos(cx_angle(k, R)));
    constexpr float s = float(cx_sin(cx_angle(k, R)));
    // (x + iy)(c - is)
    return c32{v.x * c + v.y * s, v.y * c - v.x * s};
  }
}

// In-place radix-R DIF on v[BASE .. BASE+R): afterwards X[k] = v[BASE + bitrev(k, log2 R)].
template <int R, int BASE, int TOT>
__device__ __forceinline__ void dif(c32 (&v)[TOT]) {
  if constexpr (R >= 2) {
    constexpr int h = R / 2;
    static_for<0, h>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const c32 a = v[BASE + i], b = v[BASE + i + h];
      v[BASE + i] = cadd(a, b);
      v[BASE + i + h] = mul_w<i, R>(csub(a, b));
    });
    dif<h, BASE, TOT>(v);
    dif<h, BASE + h, TOT>(v);
  }
}

// ---- decimation-in-time butterflies with the twiddle folded into FMAs ---------------------------
// (e, o) -> (e + w o, e - w o), w = exp(-2 pi i K / R) a compile-time constant.  A general twiddle costs
// 6 instructions (4 literal-operand FMAs for e + w o, then 2 e - that) against 8 for "multiply, add, subtract".
// EXACT: the difference output is formed as e - w o with its own two FMAs instead of 2 e - (e + w o) - one
// rounding less where the two operands cancel (used in the last stage of a frame, where the 31 other bins
// of a strong tone's butterfly must come out near zero).
template <int K, int R, bool EXACT = false>
__device__ __forceinline__ void bf_w(c32& e, c32& o) {
  constexpr int k = ((K % R) + R) % R;
  constexpr float s8 = 0.70710678118654752440f;
  if constexpr (k == 0) {
    const c32 a = e, b = o;
    e = cadd(a, b); o = csub(a, b);
  } else if constexpr (4 * k == R) {            // w = -i : w o = (o.y, -o.x)
    const c32 a = e, b = o;
    e = c32{a.x + b.y, a.y - b.x}; o = c32{a.x - b.y, a.y + b.x};
  } else if constexpr (2 * k == R) {            // w = -1
    const c32 a = e, b = o;
    e = csub(a, b); o = cadd(a, b);
  } else if constexpr (4 * k == 3 * R) {        // w = +i : w o = (-o.y, o.x)
    const c32 a = e, b = o;
    e = c32{a.x - b.y, a.y + b.x}; o = c32{a.x + b.y, a.y - b.x};
  } else if constexpr ((8 * k) % R == 0) {      // odd multiples of R/8: w = (+-1 +- i)/sqrt2
    const float p = o.x + o.y, m = o.y - o.x;   // (1 - i)/sqrt2 * o = s8 (p + i m)
    float tr, ti;                                // t = w o / s8
    if constexpr (8 * k == R) { tr = p; ti = m; }
    else if constexpr (8 * k == 3 * R) { tr = m; ti = -p; }
    else if constexpr (8 * k == 5 * R) { tr = -p; ti = -m; }
    else { tr = -m; ti = p; }
    const c32 x1 = c32{fmaf(s8, tr, e.x), fmaf(s8, ti, e.y)};
    if constexpr (EXACT) o = c32{fmaf(-s8, tr, e.x), fmaf(-s8, ti, e.y)};
    else o = c32{fmaf(2.0f, e.x, -x1.x), fmaf(2.0f, e.y, -x1.y)};
    e = x1;
  } else {
    constexpr float c = float(cx_cos(cx_angle(k, R)));
    constexpr float s = float(cx_sin(cx_angle(k, R)));   // w = c - i s :  w o = (o.x c + o.y s) + i (o.y c - o.x s)
    const c32 x1 = c32{fmaf(o.y, s, fmaf(o.x, c, e.x)), fmaf(-o.x, s, fmaf(o.y, c, e.y))};
    if constexpr (EXACT) o = c32{fmaf(-o.y, s, fmaf(-o.x, c, e.x)), fmaf(o.x, s, fmaf(-o.y, c, e.y))};
    else o = c32{fmaf(2.0f, e.x, -x1.x), fmaf(2.0f, e.y, -x1.y)};
    e = x1;
  }
}

// In-place radix-R DIT on the registers v[BASE + i*STRIDE], i < R (natural input order): afterwards
// X[k] = v[BASE + STRIDE * bitrev(k, log2 R)] - the same output convention as dif<>.
// SKIP_FIRST: the first layer (the R == 2 leaves: pairs b, b + TOT_R/2) has already been done by the caller
// (bf_tw below, fused with the pass's pre-twiddle)
template <int R, int BASE, int STRIDE, int TOT, bool SKIP_FIRST = false>
__device__ __forceinline__ void dit_s(c32 (&v)[TOT]) {
  if constexpr (R == 2) {
    if constexpr (!SKIP_FIRST) bf_w<0, 2>(v[BASE], v[BASE + STRIDE]);
  } else if constexpr (R > 2) {
    constexpr int h = R / 2, L1 = ilog2(h);
    dit_s<h, BASE, 2 * STRIDE, TOT, SKIP_FIRST>(v);              // E = DFT of the even samples
    dit_s<h, BASE + STRIDE, 2 * STRIDE, TOT, SKIP_FIRST>(v);     // O = DFT of the odd samples
    static_for<0, h>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int slot = BASE + 2 * STRIDE * bitrev(k, L1);       // E[k]; O[k] sits STRIDE above
      bf_w<k, R>(v[slot], v[slot + STRIDE]);
    });
  }
}
template <int R, int BASE, int TOT>
__device__ __forceinline__ void dit(c32 (&v)[TOT]) { dit_s<R, BASE, 1, TOT>(v); }
// the same network without its first layer
template <int R, int BASE, int TOT>
__device__ __forceinline__ void dit_rest(c32 (&v)[TOT]) { dit_s<R, BASE, 1, TOT, true>(v); }

// First layer of a pass fused with the pass's pre-twiddle:  (e, o) -> (e te + o to, e te - o to)  in ten
// instructions (cmul for e te, four FMAs for the sum, the difference as 2 e te - sum) instead of twelve
// (two complex multiplies, add, subtract)
__device__ __forceinline__ void bf_tw(c32& e, c32& o, c32 te, c32 to) {
  const c32 et = cmul(e, te);
  const float x1 = fmaf(to.x, o.x, fmaf(-to.y, o.y, et.x));
  const float y1 = fmaf(to.x, o.y, fmaf(to.y, o.x, et.y));
  o = c32{fmaf(2.0f, et.x, -x1), fmaf(2.0f, et.y, -y1)};
  e = c32{x1, y1};
}

// Same butterfly network, depth first, calling emit(integral_constant<register index>) as soon as a
// register holds a final output: the LDS stores of the first outputs then issue while the rest of the
// butterfly is still being computed (X[k] sits in register BASE + bitrev(k)).
template <int R, int BASE, int TOT, class F>
__device__ __forceinline__ void dif_emit(c32 (&v)[TOT], F&& emit) {
  if constexpr (R == 1) {
    emit(std::integral_constant<int, BASE>{});
  } else {
    constexpr int h = R / 2;
    static_for<0, h>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const c32 a = v[BASE + i], b = v[BASE + i + h];
      v[BASE + i] = cadd(a, b);
      v[BASE + i + h] = mul_w<i, R>(csub(a, b));
    });
    dif_emit<h, BASE, TOT>(v, emit);
    dif_emit<h, BASE + h, TOT>(v, emit);
  }
}

// Multiply v[c] by w^c for c = 1..31 where w^1..w^3 (lo) and w^4, w^8 .. w^28 (hi) come exact from
// the twiddle table: every factor is at most ONE rounded product away from the table value.
__device__ __forceinline__ void twiddle32(c32 (&v)[32], const c32 (&lo)[3], const c32 (&hi)[7]) {
  static_for<1, 32>([&](auto ic) {
    constexpr int c = decltype(ic)::value;
    constexpr int h = c >> 2, l = c & 3;
    if constexpr (h == 0) {
      v[c] = cmul(v[c], lo[l - 1]);
    } else if constexpr (l == 0) {
      v[c] = cmul(v[c], hi[h - 1]);
    } else {
      v[c] = cmul(v[c], cmul(hi[h - 1], lo[l - 1]));
    }
  });
}

}  // namespace tdsa
