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
