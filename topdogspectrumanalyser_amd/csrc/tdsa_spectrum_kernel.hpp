// tdsa_spectrum_kernel.hpp - the fused frame kernel: one LDS-resident FFT frame per workgroup slot.
//
//   raw int8/uint8/complex64 IQ  ->  (x - dc) * window  ->  N-point FFT (A x 32 x 32 Stockham, data
//   exchanged through LDS between register-resident radix passes)  ->  fftshift  ->  |X| / |X|^2  ->
//   dB (+floor, +cal offset, -tare)  ->  per-frame dB row + register-resident max/min hold.
//
// Replaces, per frame, datasources/hackrf_samples.py:357-383 and datasources/rtl_samples.py:167-184
// of the reference (numpy mean / window multiply / np.fft.fft / fftshift / abs / log10) and
// core/display_data_processor.py:177-181 (cal offset, hold).  HBM traffic per frame is the
// algorithmic minimum: the raw samples are read once, the dB row is written once.
//
// Work decomposition (N = A * 32 * 32 for N >= 2048, N = A * 32 below):
//   * 2 * N/32 threads own one frame: N/32 butterfly rows, each shared by two half-threads (lanes l and l + 32 of a
//     wave) that hold 16 complex points apiece in every pass - 128 VGPRs, 4 waves per SIMD.  A radix-R pass is two
//     radix-R/2 DFTs (one per half-thread) plus one combine stage across the lane pair (v_permlane32_swap).
//   * pass 1: M = 32/A adjacent radix-A butterflies per row, inputs straight from global memory (row t reads samples
//             a*(N/A) + t*M .. +M-1 : 2M contiguous bytes per load), no twiddles.  The raw registers are refilled
//             with the NEXT frame's samples as soon as they have been unpacked, so the HBM read of frame f+1
//             overlaps the FFT of frame f at no register cost.
//   * pass 2 (3-pass sizes): radix-32 IN PLACE (a row stores output kb into the LDS slot input b = kb came from, so
//             no barrier separates its gather from its scatter), twiddles W_(32A)^(ka*b) read from a 4 KiB LDS
//             table.  Outputs are stored depth-first, as soon as they are final, so the slow LDS write path
//             (~93 B/clk/CU) drains underneath the remaining butterflies.
//   * last pass: radix-32, twiddles W_N^(t*c) rebuilt from 7 exact per-thread table values (every factor is at most
//             one rounded product away from the table).
//   * the thread -> bin mapping of the last pass is frame invariant, so the window slice, the twiddle seeds and the
//     max/min hold traces live in registers across the persistent frame loop.
//   * a workgroup processes a CONTIGUOUS range of frames (overlapping frames re-read their shared half from the
//     same XCD's L2, not from HBM); the dB rows leave with non-temporal stores.
#pragma once
#ifndef TDSA_DEV          // the two developer builds left (tools/build_variants.sh dev "-DTDSA_DEV -DTDSA_TIMELINE"):
#undef TDSA_TIMELINE      //   s_memtime stamps of workgroup 0 at the phase boundaries (tdsa_debug_timeline, tools/timeline.py)
#undef TDSA_DYN_PRIO      //   falling issue priorities through a barrier phase (profiles/r03_c3_timeline_dynprio.txt)
#endif
#include "tdsa_fft.hpp"
#include "tdsa_kernels.hpp"

namespace tdsa {

template <int LOG2N>
struct Cfg {
  static constexpr int N = 1 << LOG2N;
  static constexpr int SG = N / 32;                       // butterfly rows per frame
  static constexpr int TPF = 2 * SG;                      // threads per frame: two half-threads per row
  static constexpr int NPASS = (N >= 2048) ? 3 : 2;
  static constexpr int A = (NPASS == 3) ? N / 1024 : N / 32;  // radix of the first pass
  static constexpr int M = 32 / A;                        // adjacent first-pass butterflies per row
  static constexpr int H = A / 2;                         // first-pass radix done inside one lane (even / odd row split)
  // INL: pass 1 entirely inside a lane (see the kernel); R1 = in-lane radix of pass 1, CPT = adjacent columns
  // (samples) per row read of a half-thread
  static constexpr bool INL = (M >= 8) && (NPASS == 3);
  static constexpr int R1 = INL ? A : H;
  static constexpr int CPT = INL ? M / 2 : M;
  static constexpr int WGT = TPF > 256 ? TPF : 256;       // threads per workgroup
  static constexpr int FPW = WGT / TPF;                   // frames in flight per workgroup
  static constexpr int NPAD = N + N / 32;                 // LDS slot: rows of 32 complex + 1 pad element
  static constexpr int WPF = SG >= 32 ? SG / 32 : 1;      // waves per frame
  static constexpr int NWAVE = WGT / 64;
  static constexpr int TWM = (NPASS == 3) ? 32 * A : 0;   // middle-pass twiddle table entries [b][ka]
  static constexpr size_t DATA_BYTES = size_t(FPW) * NPAD * sizeof(c32);
  // Frames that live inside one wave (N <= 1024) run without workgroup barriers, so nothing absorbs the
  // wait for the previous frame's stores that a global window load behind them would pick up (gfx9 counts
  // loads and stores in one in-order vmcnt): those sizes keep the window table in LDS instead.
  static constexpr bool WIN_LDS = TPF <= 64;
  static constexpr size_t WIN_BYTES = WIN_LDS ? size_t(N) * sizeof(float) : 0;
  // N = 1024: room for the last pass's seven per-thread twiddles (max + min hold instantiation of the byte formats: its
  // loop has no 14 VGPRs to keep them in) - 3 x SG + 4 x TPF entries; with it four workgroups still fit a CU exactly
  static constexpr size_t TWF_BYTES = (LOG2N == 10) ? size_t(3 * SG + 4 * TPF) * sizeof(c32) : 0;
  static constexpr size_t LDS_BYTES = DATA_BYTES + size_t(TWM) * sizeof(c32) + NWAVE * 2 * sizeof(double) + TWF_BYTES + WIN_BYTES;
  static constexpr size_t LDS_ALLOC = LDS_BYTES
#ifdef TDSA_TIMELINE
      + 16 * 8 * 16 * 8
#endif
      ;
};

// unaligned-tolerant wide loads (frame starts are only guaranteed to be sample (2 byte) aligned)
struct __attribute__((packed, aligned(2))) PU1 { uint32_t x; };
struct __attribute__((packed, aligned(2))) PU2 { uint32_t x, y; };
struct __attribute__((packed, aligned(2))) PU4 { uint32_t x, y, z, w; };

template <int DW>
__device__ __forceinline__ void load_raw(const unsigned char* q, uint32_t* dst) {
  if constexpr (DW == 1) {
    dst[0] = reinterpret_cast<const PU1*>(q)->x;
  } else if constexpr (DW == 2) {
    const PU2 r = *reinterpret_cast<const PU2*>(q);
    dst[0] = r.x; dst[1] = r.y;
  } else {
    static_for<0, DW / 4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const PU4 r = *reinterpret_cast<const PU4*>(q + 16 * i);
      dst[4 * i] = r.x; dst[4 * i + 1] = r.y; dst[4 * i + 2] = r.z; dst[4 * i + 3] = r.w;
    });
  }
}

template <int W, class T>
__device__ __forceinline__ T seg_sum(T x) {
#pragma unroll
  for (int off = W / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

// keeps a value opaque to loop-invariant code motion (the twiddle products must be rebuilt per frame,
// hoisting them would cost 42 VGPRs per pass)
__device__ __forceinline__ void opaque(c32& w) { asm volatile("" : "+v"(w.x), "+v"(w.y)); }
// pins a wave-uniform float in a VGPR: VALU ops with an SGPR source issue at half rate on gfx950
__device__ __forceinline__ float in_vgpr(float x) { asm volatile("" : "+v"(x)); return x; }

// v_max_f32 / v_min_f32 as ONE instruction (np.fmax / np.fmin semantics: a NaN operand is ignored);
// the compiler's fmaxf first canonicalises both inputs with two extra v_max (half-rate ops on gfx950)
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// wave64 float maximum (NaN operands ignored) / sum across the lanes; the result is valid in lane 63
__device__ __forceinline__ float dpp_wave_max63(float x) {
  // v_max_f32 with the DPP operand in ONE instruction each (through the builtins it is v_mov, v_mov_dpp, v_max and three
  // s_nop per step); the two wait states a DPP read of a freshly written VGPR needs sit inside every statement, whatever
  // the compiler puts in front of it (its hazard recognizer does not look into inline assembly)
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x));
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0" : "+v"(x));
  return x;
}
__device__ __forceinline__ float dpp_wave_sum63(float x) {
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0xB1, 0xf, 0xf, true));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x4E, 0xf, 0xf, true));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x141, 0xf, 0xf, true));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x140, 0xf, 0xf, true));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x142, 0xa, 0xf, false));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x143, 0xc, 0xf, false));
  return x;
}

// wave64 integer sum with DPP adds (no LDS round trips); the total is valid in lanes 48..63
__device__ __forceinline__ int dpp_wave_sum(int x) {
  // bound_ctrl: every source lane of these four steps exists and the masks are full, so `old` is never read
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);    // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);    // row_mirror: every lane = row sum
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return x;
}

// float max/min through integer atomics (IEEE-754 order trick); NaN candidates are skipped (np.fmax)
__device__ __forceinline__ void atomic_fmax_dev(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else if (v < 0.f) atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_fmin_dev(float* addr, float v) {
  if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else if (v < 0.f) atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(bytes), 0x00020000);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// NB consecutive dwords starting at byte (voff + soff) of the buffer
template <int NB>
__device__ __forceinline__ void buf_load(rsrc_t r, unsigned voff, unsigned soff, uint32_t* dst) {
  if constexpr (NB == 1) {
    dst[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
  } else if constexpr (NB == 2) {
    const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    dst[0] = q.x; dst[1] = q.y;
  } else {
    static_for<0, NB / 4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + 16 * i, 0);
      dst[4 * i] = q.x; dst[4 * i + 1] = q.y; dst[4 * i + 2] = q.z; dst[4 * i + 3] = q.w;
    });
  }
}

// cache policy of the dB row stores: non-temporal (gfx940+ "nt" bit).  The rows are written once and not read again
// by this kernel; streaming them keeps the input's shared halves in L2 and the Infinity Cache free of 160 MB of
// write-once data per launch: -4 % per launch at C3, -11 % at C2 once the working set exceeds the 256 MiB cache
// (bench.py's ring of 8 buffers), no change when everything fits (profiles/r02_c3_experiments.txt)
constexpr int kRowStorePolicy = 2;
constexpr float k10Log10_2 = 3.01029995663981195214f;   // 10*log10(2)
constexpr float kMagExactBelow = 1e-8f;                  // |X|^2 below this: the 1e-12 floor is visible

// (The timing-only ablation switches of rounds 2-4 - no barriers / LDS exchange / stores / swaps / log / twiddles, fewer
//  waves, the DIF network, unfused twiddles and window, staggered workgroups - went out with round 5: what they measured is
//  in profiles/r03_c3_ablation.txt, r03_c3_experiments.txt, r04_c3_proto.txt and profiles/HISTORY.md; the tree at the
//  round-4 verdict (git: 2ece423) still builds them.  tests/test_isa_frozen.py pins the instruction streams they
//  were taken out of.)
// Phase boundary inside the frame loop.  When a frame lives inside one wave (N <= 1024) everything the
// phases exchange through LDS is private to that wave: its DS operations execute in order, so only the
// compiler has to be kept from reordering across the boundary and the waves of a workgroup run free of
// each other.  Larger frames span several waves and need the workgroup barrier.
#define TDSA_SYNC()                                                        \
  do {                                                                     \
    if constexpr (C::TPF <= 64) {                                          \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               \
      __builtin_amdgcn_wave_barrier();                                     \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");               \
    } else {                                                               \
      __syncthreads();                                                     \
    }                                                                      \
  } while (0)

// Issue priority that falls as a wave advances through a barrier phase (s_setprio 3 .. 0 at the quarter marks).  The
// SIMD arbiter serves the highest priority first and the oldest wave among equals: with one flat priority the oldest
// wave of a SIMD races through every phase and then idles at the barrier while the youngest finishes the phase alone,
// at the rate ONE wave can issue (about half of what the SIMD sustains).  With falling priorities a wave that is ahead
// yields to the ones behind it, the four waves of a SIMD reach the barrier together and the SIMD stays saturated.
#ifdef TDSA_DYN_PRIO
#define TDSA_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define TDSA_PRIO(n) do { } while (0)
#endif

#ifdef TDSA_TIMELINE
// developer build: wave 0..7 of workgroup 0 stamp s_memtime into LDS at phase boundaries (first 8 frames)
#define TDSA_STAMP(i)                                                                              \
  do {                                                                                             \
    if (p.dbg != nullptr && blockIdx.x == 0 && (tid & 63) == 0 && unit - u0 < 8)                   \
      tl[((unit - u0) * 16 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime();                      \
  } while (0)
#else
#define TDSA_STAMP(i)
#endif

// LDS data accesses go through volatile 64-bit vectors: that keeps the backend from pairing them into
// ds_read2_b64 / ds_write2_b64, which the gfx950 LDS pipe serves at 0.3 TB/s/CU against 0.53 TB/s/CU for
// plain ds_read_b64 (tools/ubench/lds_rate.hip), while leaving waitcnt bookkeeping to the compiler.
typedef float lds_v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile lds_v2* lds_vptr;
__device__ __forceinline__ c32 lds_ld(const c32* p) {
  const lds_v2 x = *(lds_vptr)(p);
  return c32{x.x, x.y};
}
__device__ __forceinline__ void lds_st(c32* p, c32 v) { *(lds_vptr)(p) = lds_v2{v.x, v.y}; }

// exchange the upper half-wave of `a` with the lower half-wave of `b` (one v_permlane32_swap per dword):
// afterwards a = [a.lo | b.lo], b = [a.hi | b.hi]
__device__ __forceinline__ void swap_halves(c32& a, c32& b) {
  auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a.x = __uint_as_float(rx[0]); b.x = __uint_as_float(rx[1]);
  a.y = __uint_as_float(ry[0]); b.y = __uint_as_float(ry[1]);
}

// In-register radix network of the passes: decimation in time with the twiddles folded into FMAs
// (tdsa_fft.hpp: 6 instructions per general butterfly instead of 8).
template <int R, int BASE, int TOT>
__device__ __forceinline__ void radix(c32 (&v)[TOT]) {
  dit<R, BASE, TOT>(v);
}

// radix-2 combine  (E, O) -> (E + w O, E - w O)  in six FMAs (second output as 2E - first)
__device__ __forceinline__ void combine(c32& e, c32& o, c32 w) {
  const float x1 = fmaf(w.x, o.x, fmaf(-w.y, o.y, e.x));
  const float y1 = fmaf(w.x, o.y, fmaf(w.y, o.x, e.y));
  o = c32{fmaf(2.0f, e.x, -x1), fmaf(2.0f, e.y, -y1)};
  e = c32{x1, y1};
}
// radix-32 combine for unit u + 8h: the lower half-wave uses W_32^u, the upper W_32^(u+8) = -i W_32^u.
// The -i is applied to O with two selects (-i (x + iy) = y - ix), the rest is compile-time constants.
template <int U, bool EXACT = false>
__device__ __forceinline__ void combine32(c32& e, c32& o, bool odd_half) {
  const c32 orot = c32{odd_half ? o.y : o.x, odd_half ? -o.x : o.y};
  o = orot;
  bf_w<U, 32, EXACT>(e, o);
}
template <int K, int R, bool EXACT = false>   // compile-time twiddle W_R^K
__device__ __forceinline__ void combine_const(c32& e, c32& o) {
  bf_w<K, R, EXACT>(e, o);
}

// Thread layout: a wave owns 32 consecutive butterfly rows; lane l < 32 is the EVEN half-thread of row
// (l & 31), lane l + 32 the ODD half-thread of the same row.  Every radix-R pass is done as two
// radix-R/2 DFTs (one per half-thread, on the even / odd indexed inputs) followed by one radix-2
// combine stage whose operands are brought together with v_permlane32_swap: after the swap the lower
// half-wave holds (E_u, O_u) for units u = 0..7 and the upper half-wave (E_(u+8), O_(u+8)), so both
// halves do full butterflies and no lane idles.  16 points per thread keeps the kernel under 128 VGPRs:
// 4 waves per SIMD, which is what it takes to keep the VALU fed (one wave issues at most one VALU op
// every 4 clocks; the SIMD retires one every 2).
// CHIRP (complex64 input, no hold, frames of whole waves only): 3 = the whole chirp-z convolution of a frame in one pass
// through the workgroup (raw samples in, dB / power rows of the N wanted bins out); 1 = its first transform alone (raw samples
// unpacked, DC-freed and multiplied by window x chirp on load; conj(X B) stored), 2 = its second transform alone (dB / power
// rows stored); 4 = the two row passes of a LONG chirp-z frame's transforms in one (16384-point rows: transform, x the
// filter spectrum's row, conjugate, through LDS, transform; complex64 rows in and out): tdsa_chirp.hip's two element-wise passes folded into the transforms, in instantiations of their own -
// as run-time branches of the plain complex64 kernel they cost it 68 - 83 spilled registers
// STATS (frames of whole waves, dB rows, HOLD 0 / 1): every wave also leaves what the per-frame scalars of the trace analytics
// need from ITS bins while they sit in its registers - maximum dB, its first bin, the band's linear power - as one 16-byte
// record per wave and frame (SpecParams::stats_part); frame_stats_finish_kernel (tdsa_analytics.hip) folds a frame's records.
// What np.max / np.argmax (core/duty_cycle.py:36, core/marker_manager.py:97) and MarkerManager._band_power (:308-319) would
// otherwise take a second pass over the rows for (rows_stats_kernel).
// (HOLDX = HOLD | 8 x STATS: one template argument, so that the names of the instantiations without STATS - what the
// profiles, tests/test_isa_frozen.py and tools/ key on - stay what they were)
template <int LOG2N, bool IN_C64, int HOLDX, int CHIRP = 0>   // HOLD: bit0 = max trace, bit1 = min trace; 4 = AGG (see below); bit 3: STATS
__global__ void __launch_bounds__(Cfg<LOG2N>::WGT, 4) spectrum_kernel(const SpecParams p) {
  constexpr int HOLD = HOLDX & 7;
  constexpr bool STATS = (HOLDX & 8) != 0;
  using C = Cfg<LOG2N>;
  constexpr int N = C::N, SG = C::SG, A = C::A, M = C::M, H = C::H, FPW = C::FPW, NPAD = C::NPAD;
  constexpr int LH = ilog2(H);
  // Pass 1 of a row = M adjacent radix-A butterflies.  INL: each half-thread takes M/2 of them WHOLE - all A input
  // rows of its M/2 adjacent columns - and pass 1 stays inside the lane (no half-thread exchange: 16
  // v_permlane32_swap less per thread and frame).  It costs twice the row reads of half the width, which is why it
  // only pays where a row read stays >= 8 bytes: same-box A/B (profiles/r03_c3_experiments.txt) N = 4096 -2.6 %,
  // 2048 -2.3 %, but 8192 (4-byte reads) +1.4 %, 256 +2.9 %, 128 +-0.  M = 2 (N = 16384, 512) would need 2-byte
  // loads into twice the registers (gfx950's d16 loads do not preserve the other half with SRAM ECC on), M = 1 is
  // one radix-32 butterfly per row: all of those keep the even / odd row split with the swap + combine stage.
  constexpr bool INL = C::INL;
  constexpr int R1 = C::R1;                       // radix done inside one lane in pass 1
  constexpr int LR1 = ilog2(R1);
  constexpr int CPT = C::CPT;                     // adjacent columns (samples) per row read of a half-thread
  constexpr int DW = (CPT >= 2) ? CPT / 2 : 1;    // raw dwords per row read
  constexpr int NRAW = IN_C64 ? 1 : R1 * DW;      // row reads r < R1: input row a = r (INL) or 2r + h
  constexpr int ROWB = (N / A) * 2;               // bytes per first-pass row (byte formats)
  constexpr int RSTEP = INL ? 1 : 2;              // input rows between two row reads of a half-thread

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c32* lds = reinterpret_cast<c32*>(smem);
  c32* twm = reinterpret_cast<c32*>(smem + C::DATA_BYTES);
  double* red = reinterpret_cast<double*>(smem + C::DATA_BYTES + size_t(C::TWM) * sizeof(c32));
#ifdef TDSA_TIMELINE
  unsigned long long* tl = reinterpret_cast<unsigned long long*>(smem + C::LDS_BYTES);
#endif

  const int tid = threadIdx.x;
  const int wave = tid >> 6;
#ifdef TDSA_TIMELINE
  const unsigned long long tl_entry = __builtin_amdgcn_s_memtime();     // launch-level stamps of workgroup 0
#endif
  const int h = (tid >> 5) & 1;                              // 0: even half-thread, 1: odd half-thread
  const int g = wave * 32 + (tid & 31);                      // row inside the workgroup
  // UNI: a frame is made of whole waves (N >= 1024), so its slot - and with it the frame index, the frame's base
  // addresses and every "is this frame there" test - is wave-uniform: taken through readfirstlane it lives in SGPRs,
  // the loads and row stores go through SGPR buffer descriptors and no per-lane 64-bit pointer is left (at N = 1024 /
  // 2048, where several frames share a workgroup, those pointers and their v_add_co / v_addc pairs cost registers the
  // max + min hold instantiation does not have)
  constexpr bool UNI = C::TPF >= 64;
  const int slot = (FPW == 1) ? 0 : (UNI ? __builtin_amdgcn_readfirstlane(g / SG) : g / SG);
  const int t = (FPW == 1) ? g : g - slot * SG;              // butterfly row inside the frame
  c32* buf = lds + slot * NPAD;

  const int n_units = (p.n_frames + FPW - 1) / FPW;
  // unit range of this workgroup: [floor(b n / g), floor((b + 1) n / g)) with n = q g + r evaluated as
  // b q + floor(b r / g) - 32-bit divisions (the 64-bit form cost two emulated 64-bit divisions at the head of
  // every launch, ahead of the first load).  The workgroups that take one unit more stay spread over the grid:
  // handing them out as one contiguous block instead measured +0.6 us per serial C3 launch.
  const unsigned upw = unsigned(n_units) / gridDim.x, urem = unsigned(n_units) - upw * gridDim.x;
  int u0 = int(blockIdx.x * upw + blockIdx.x * urem / gridDim.x);
  int u1 = int((blockIdx.x + 1) * upw + (blockIdx.x + 1) * urem / gridDim.x);
  // frame of this slot in workgroup-unit `unit`: every FPW-th frame of the workgroup's range - or, where the averager's
  // chunk aggregate is formed (HOLD == 4), the slot's own run of consecutive frames
  auto frame_of = [&](int unit) -> int {
    if constexpr (HOLD == 4 && FPW > 1) return u0 * FPW + slot * (u1 - u0) + (unit - u0);
    else return unit * FPW + slot;
  };
  constexpr unsigned SB = IN_C64 ? 8u : 2u;                      // bytes per sample
  // ---- frame-invariant per-thread state ---------------------------------------------------------
  const rsrc_t win_rsrc = make_rsrc(C::WIN_LDS ? p.window : p.window_perm, N * 4u);
  float* win_lds = reinterpret_cast<float*>(smem + C::LDS_BYTES - C::WIN_BYTES);
  // TWF_LDS: the last pass's per-thread twiddles W_N^(t (8a + h)), W_N^(2 t j) are re-read from LDS every frame instead
  // of living in 14 VGPRs (see Cfg::TWF_BYTES)
  constexpr bool TWF_LDS = C::TWF_BYTES != 0 && HOLD == 3 && !IN_C64;
  c32* twf_tab = reinterpret_cast<c32*>(smem + C::LDS_BYTES - C::WIN_BYTES - C::TWF_BYTES);    // [3][SG] lo, then [4][TPF] hi
  if constexpr (TWF_LDS) {
    if (slot == 0) {
      if (h == 0) static_for<0, 3>([&](auto ic) { constexpr int j = decltype(ic)::value; twf_tab[j * SG + t] = p.tw[t * 2 * (j + 1)]; });
      static_for<0, 4>([&](auto ic) { constexpr int a = decltype(ic)::value; twf_tab[3 * SG + a * C::TPF + h * SG + t] = p.tw[t * (8 * a + h)]; });
    }
  }
  if constexpr (C::WIN_LDS) {
    for (int i = tid; i < N; i += C::WGT) win_lds[i] = p.window[i];
    __syncthreads();
  }
  // the thread's 16 window values: quarter q of them are the 16 bytes at ((q * 2 SG + h SG + t) * 16 of the permuted
  // table (window_perm_kernel below): every load instruction of a wave covers 2 x 512 contiguous bytes
  const unsigned win_voff = (unsigned(h) * SG + unsigned(t)) * 16u;
  const bool odd_half = h != 0;   // upper half-wave: its radix-32 combine twiddle is W_32^(u+8) = -i * W_32^u
  // AGG (HOLD == 4: linear-power rows for the TraceAverager, one frame per workgroup slot): the workgroup also forms the
  // chunk aggregate the chained scan of tdsa_trace.hip needs for ITS frames [u0, u1) - the averager's recurrence
  // s <- a_f s + b_f P_f (utils/signal_processing.py:35-61) run from a zero state, which is the dot product
  // L = sum_f w_f P_f with w_f = b_f prod_(g > f) a_g, the same weights for every bin (SpecParams::agg_w, made in float64
  // and rounded once) - and leaves it as its row of agg_out.  Until round 4 that took a pass of its own over the rows.
  // float32: sixteen registers are what this instantiation has to spare (float64 sums spilled 13 dwords); the aggregate
  // only carries what the chunk's ~10 frames ADD to the state - its rounding (<= 2^-24 per term) enters the float64
  // chain scaled by L / s <= 1 and decays with the chain's multipliers, while the chunks are re-scanned in float64.
  constexpr bool AGG = HOLD == 4;
  // Several frames per workgroup (N <= 2048): slot s takes the s-th run of consecutive frames of the workgroup's range
  // (frame_of below) instead of every FPW-th one, the weights are those of the workgroup's whole range, and the slots'
  // partial sums are added through LDS at the end - one aggregate row per workgroup as at the larger sizes.
  float agg[AGG ? 16 : 1];
  static_for<0, (AGG ? 16 : 1)>([&](auto ic) { agg[decltype(ic)::value] = 0.f; });
  float hmax[(HOLD & 1) ? 16 : 1], hmin[(HOLD & 2) ? 16 : 1];
  static_for<0, 16>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr ((HOLD & 1) != 0) hmax[i] = -INFINITY;
    if constexpr ((HOLD & 2) != 0) hmin[i] = INFINITY;
  });
  // STATS: which of the thread's 16 bins (display position (kcs + 8h) SG + t) lie in the band [band_lo, band_hi] is a
  // property of the HALF-WAVE (32 consecutive positions) for all but the two runs that hold the band's ends: bit q of
  // st_mask = the run of bin q in lanes 0 - 31 lies inside the band, bit 16 + q = the run in lanes 32 - 63 does; st_part,
  // same layout = the run is cut by an end of the band (at most two runs of a frame: their waves test every lane).
  unsigned st_mask = 0u, st_part = 0u;
  if constexpr (STATS) {
    static_assert(UNI && CHIRP == 0 && HOLD < 2, "frame statistics: whole waves per frame, plain dB rows, no min hold");
    const int st_t0 = __builtin_amdgcn_readfirstlane((wave & (C::WPF - 1)) * 32);     // first butterfly row of the wave
    const int lo = int(p.band_lohi & 0xffffu), hi = int(p.band_lohi >> 16);
    static_for<0, 16>([&](auto ic) {
      constexpr int q = decltype(ic)::value;
      constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
      const int a0 = kcs * SG + st_t0, a1 = a0 + 8 * SG;           // first position of the lower / upper half-wave
      const bool in0 = lo <= a0 && a0 + 31 <= hi, in1 = lo <= a1 && a1 + 31 <= hi;
      const bool out0 = hi < a0 || lo > a0 + 31, out1 = hi < a1 || lo > a1 + 31;
      const bool cut0 = lo <= hi && !in0 && !out0, cut1 = lo <= hi && !in1 && !out1;   // (lo > hi: no band)
      st_mask |= (unsigned(in0) << q) | (unsigned(in1) << (16 + q));
      st_part |= (unsigned(cut0) << q) | (unsigned(cut1) << (16 + q));
    });
    st_mask = __builtin_amdgcn_readfirstlane(st_mask);
    st_part = __builtin_amdgcn_readfirstlane(st_part);
  }
  // (with max AND min hold in registers the loop has no VGPR to spare: the wave-uniform constants stay in SGPRs there
  //  and their few uses issue at half rate)
  constexpr bool PIN = HOLD < 3;
  const unsigned xm_v = [&] { unsigned x = (M == 1) ? (p.xor_mask & 0xffffu) : p.xor_mask; if constexpr (PIN) asm volatile("" : "+v"(x)); return x; }();
  // epilogue constants in VGPRs.  DB_MAG is evaluated as 10*log10(|X|^2): identical to
  // 20*log10(|X| + 1e-12) in float32 whenever |X|^2 >= 1e-8 (the floor is below half an ulp of |X|);
  // frames with a smaller bin take the exact path below.
  const bool mag_mode = p.db_mode == 0;
  const float ps_v = PIN ? in_vgpr(mag_mode ? 1.0f : p.pscale) : (mag_mode ? 1.0f : p.pscale);
  const float fl_v = PIN ? in_vgpr(mag_mode ? 0.0f : p.log_floor) : (mag_mode ? 0.0f : p.log_floor);
  const float cal_v = PIN ? in_vgpr(p.cal_db) : p.cal_db;

  // LDS addressing in complex elements: element i of a pass lives at i + (i >> 5)
  const int wr1_base = 33 * t + (A == 32 ? 8 : 16) * h;          // pass 1 writes row t (INL: CPT * A = 16 elements per half)
  constexpr int rd_stride = SG + SG / 32;
  const int rd_base = (SG % 32 == 0) ? t + (t >> 5) : 0;         // gather of y[t + b*SG]
  const int rdA = rd_base + h * rd_stride;                       // element b = 2i + h
  const int wrM = rd_base + 8 * h * rd_stride;                   // in-place output kb = u + 8h
  const int ka_mid = t % A;
  const int rd3A = (t / A) * rd_stride + ka_mid + h * A;         // last gather, element c = 2i + h

  const unsigned lane_in_off = INL ? (unsigned(t) * M + unsigned(h) * CPT) * SB
                                   : unsigned(t) * (M * SB) + unsigned(h) * ((N / A) * SB);
  const unsigned out_voff = unsigned(t) * 4u + unsigned(h) * (8u * SG * 4u);
  float st_band = 0.f;                          // this thread's share of the frame's band power (linear)
  unsigned st_maskc = 0u;                       // st_mask as the frame at hand sees it (an opaque copy: turned into lane masks bin
                                                // by bin inside the loop, not hoisted out of it as 32 scalar registers)
  // a bin whose two runs are whole: the lane mask is two sign-extended bits (scalar unit), the add one select + one add
  auto st_band_add = [&](auto qc, float a) {
    constexpr int q = decltype(qc)::value;
    const unsigned mlo = unsigned(__builtin_amdgcn_sbfe(int(st_maskc), q, 1)), mhi = unsigned(__builtin_amdgcn_sbfe(int(st_maskc), 16 + q, 1));
    const unsigned long long m = mlo | ((unsigned long long)mhi << 32);
    float t;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(t) : "v"(a), "s"(m));
    st_band += t;
  };
  // a wave that holds an end of the band: every lane tests its own position
  auto st_band_add_exact = [&](auto qc, float a) {
    constexpr int q = decltype(qc)::value;
    constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
    unsigned ov = out_voff;                     // (opaque: the test must not leave the loop as sixteen lane masks)
    asm volatile("" : "+v"(ov));
    const unsigned lo = p.band_lohi & 0xffffu, hi = p.band_lohi >> 16;
    st_band += ((ov >> 2) + unsigned(kcs * SG) - lo <= hi - lo) ? a : 0.f;
  };
  // frame -> byte offset of its samples / element offset of its output row (several captures per launch:
  // SpecParams::seg_*; with one capture seg_magic = 0 and these are frame * frame_stride, frame * N)
  auto in_byte_off = [&](int frame) -> long long {
    const unsigned sgm = __umulhi(unsigned(frame), p.seg_magic);
    const unsigned fi = unsigned(frame) - sgm * p.seg_frames;
    return (long long)sgm * p.seg_in_stride + (long long)fi * p.frame_stride;
  };
  auto out_elem_off = [&](int frame) -> long long {
    const unsigned sgm = __umulhi(unsigned(frame), p.seg_magic);
    const unsigned fi = unsigned(frame) - sgm * p.seg_frames;
    return (long long)sgm * p.seg_out_stride + (long long)fi * N;
  };

  // Window slice of this thread (sizes that keep the table in global memory): (re)loaded at the END of a
  // frame, ahead of that frame's dB stores.  gfx9 counts loads and stores in one in-order vmcnt, so a load
  // issued behind the stores can only be waited for together with them - the waves that reach the next
  // barrier last would sit through their own stores' round trip to memory on the critical path.  The
  // reload is unconditional (the last one is simply unused) so that the old values are dead in between.
  float win[16];                       // win[c*R1 + r] = w[a(r)*(N/A) + t*M + c (+ h*CPT if INL)], a(r) = r (INL) or 2r + h
  auto load_window = [&] {
    if constexpr (CHIRP != 0) {
      // (the transforms of a chirp-z plan carry no window: it is part of the table the samples meet on load)
    } else if constexpr (C::WIN_LDS) {
      static_for<0, R1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, CPT>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          if constexpr (INL) win[jj * R1 + i] = win_lds[i * (N / A) + t * M + h * CPT + jj];
          else win[jj * R1 + i] = win_lds[(2 * i + h) * (N / A) + t * M + jj];
        });
      });
    } else {
      // four 16-byte loads (at N = 16384 the natural table order meant eight 8-byte ones: more, narrower vector
      // memory instructions cost this kernel more than their bytes - profiles/r03_c3_experiments.txt)
      uint32_t wq[16];
      static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; buf_load<4>(win_rsrc, win_voff, q * (2u * SG * 16u), &wq[4 * q]); });
      static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; win[i] = __uint_as_float(wq[i]); });
    }
  };
  uint32_t raw[NRAW];
  auto load_frame_raw = [&](int frame) {
    if constexpr (!IN_C64) {
      const bool act = frame < p.n_frames;
      const unsigned char* fb = static_cast<const unsigned char*>(p.in) + in_byte_off(frame);
      if constexpr (UNI) {   // frame is wave-uniform: SGPR descriptor + lane offset (a frame that is not there: zero
                             // records, its loads return zeros)
        const rsrc_t r = make_rsrc(fb, (FPW == 1 || act) ? N * 2u : 0u);
        static_for<0, R1>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (M == 1) raw[i] = unsigned(__builtin_amdgcn_raw_buffer_load_b16(r, lane_in_off, RSTEP * i * ROWB, 0));
          else buf_load<DW>(r, lane_in_off, RSTEP * i * ROWB, &raw[i * DW]);
        });
      } else {
        static_for<0, R1>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const unsigned char* row = fb + RSTEP * i * ROWB + lane_in_off;
          if (act) {
            if constexpr (M == 1) raw[i] = *reinterpret_cast<const uint16_t*>(row);
            else load_raw<DW>(row, &raw[i * DW]);
          } else {
            static_for<0, DW>([&](auto jc) { raw[i * DW + decltype(jc)::value] = 0u; });
          }
        });
      }
    }
  };
  if (u0 < u1) load_frame_raw(frame_of(u0));
  // order of the cold fetches: the first frame's bytes (the DC sums need them first), the twiddles (the
  // middle-pass table goes through registers into LDS, which waits for everything issued before it), the
  // window slice last (64 KiB per workgroup, not needed before the first barrier has been passed)
  c32 twf_lo[3], twf_hi[4];   // last pass: W_N^(t*(2i+h)), i = 4a + j  ->  hi[a] = W^(t(8a+h)), lo[j-1] = W^(2tj)
  // TWF_RELOAD: the one-launch chirp-z instantiations of 2048 and 16384 points have no 14 registers to keep them in across
  // the frame loop (1 / 18 spilled dwords) and the LDS no room for a table: re-read from the L2-resident table per transform
  constexpr bool TWF_RELOAD = (CHIRP == 3 || CHIRP == 4) && (LOG2N == 11 || LOG2N == 14);
  if constexpr (!TWF_LDS && !TWF_RELOAD) {
    static_for<0, 3>([&](auto ic) { constexpr int j = decltype(ic)::value; twf_lo[j] = p.tw[t * 2 * (j + 1)]; });
    static_for<0, 4>([&](auto ic) { constexpr int a = decltype(ic)::value; twf_hi[a] = p.tw[t * (8 * a + h)]; });
  }
  if constexpr (C::NPASS == 3) {   // middle pass table twm[b*A + ka] = W_(32A)^(ka*b)
    if (tid < C::TWM) {
      const int b = tid / A, ka = tid % A;
      twm[tid] = p.tw[ka * b * (N / (32 * A))];
    }
  }
  if constexpr (!C::WIN_LDS) load_window();

#ifdef TDSA_TIMELINE
  if (p.dbg != nullptr && blockIdx.x == 0 && (tid & 63) == 0) {
    tl[(2 * 16 + wave) * 16 + 12] = tl_entry;
    tl[(2 * 16 + wave) * 16 + 13] = __builtin_amdgcn_s_memtime();         // prologue done (loads waited for)
  }
#endif
  for (int unit = u0; unit < u1; ++unit) {
    const int frame = frame_of(unit);
    const bool active = frame < p.n_frames;
    TDSA_STAMP(0);

    c32 v[16];                                              // pass 1: v[jj*H + i] = sample a = 2i + h of butterfly jj
    float sub_re = p.in_off, sub_im = p.in_off;
    float res_re = 0.f, res_im = 0.f;     // DC_TRACKED: the estimate as a small residual on top of in_off (see below)
    // ---- frame sums for DC removal ---------------------------------------------------------------
    if constexpr (IN_C64) {
      if constexpr (CHIRP == 1 || CHIRP == 3) {
        static_assert(HOLD == 0 && UNI, "fused chirp transforms: complex64 instantiation without hold, whole waves per frame");
        {
          // chirp-z plans (tdsa_chirp.hip): the frame's RAW samples, unpacked, DC-freed and multiplied by window x chirp
          // on the way in - step 1 folded into this load, the rows U[f][M] are neither written nor read.  The frame is
          // wave-uniform here: SGPR descriptors, and samples from in_valid on (the implied zero padding) come back as zeros
          // from the table's descriptor, which ends there.
          const unsigned char* rb = static_cast<const unsigned char*>(p.pre_raw) + (long long)frame * p.pre_stride;
          const unsigned nv = active ? unsigned(p.in_valid) : 0u;
          const rsrc_t ar = make_rsrc(p.pre_aw, nv * 8u);
          float dcx = 0.f, dcy = 0.f;
          if (active && p.dc_sub != nullptr) { const float2 d = p.dc_sub[frame]; dcx = d.x; dcy = d.y; }
          const unsigned xm = p.pre_xor & 0xffffu;
          if (p.pre_c64) {
            const rsrc_t rr = make_rsrc(rb, nv * 8u);
            static_for<0, 16>([&](auto ic) {
              constexpr int idx = decltype(ic)::value;
              constexpr int jj = idx / R1, i = idx % R1;
              const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rr, lane_in_off + jj * 8u, RSTEP * i * (N / A) * 8, 0);
              const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(ar, lane_in_off + jj * 8u, RSTEP * i * (N / A) * 8, 0);
              v[idx] = cmul(c32{__uint_as_float(q.x) - dcx, __uint_as_float(q.y) - dcy}, c32{__uint_as_float(w.x), __uint_as_float(w.y)});
            });
          } else {
            const rsrc_t rr = make_rsrc(rb, nv * 2u);
            const unsigned roff = lane_in_off >> 2;          // 2 bytes per raw sample against 8 per complex64 one
            static_for<0, 16>([&](auto ic) {
              constexpr int idx = decltype(ic)::value;
              constexpr int jj = idx / R1, i = idx % R1;
              const unsigned u = unsigned(__builtin_amdgcn_raw_buffer_load_b16(rr, roff + jj * 2u, RSTEP * i * (N / A) * 2, 0)) ^ xm;
              const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(ar, lane_in_off + jj * 8u, RSTEP * i * (N / A) * 8, 0);
              // (x - pre_off is exact: small integers / halves; a sample past in_valid meets a zero of the table)
              v[idx] = cmul(c32{(float(u & 0xffu) - p.pre_off) - dcx, (float((u >> 8) & 0xffu) - p.pre_off) - dcy},
                            c32{__uint_as_float(w.x), __uint_as_float(w.y)});
            });
          }
        }
      } else {
      const unsigned char* fb = static_cast<const unsigned char*>(p.in) + in_byte_off(frame);
      static_for<0, 16>([&](auto ic) {
        constexpr int idx = decltype(ic)::value;
        constexpr int jj = idx / R1, i = idx % R1;
        const c32* row = reinterpret_cast<const c32*>(fb + RSTEP * i * (N / A) * 8 + lane_in_off);
        if constexpr (HOLD == 0) {
          // zero-padded rows (chirp-z plans): the padding is not read.  Sample index of this element:
          // a(i) N/A + t M + jj (+ h CPT), a(i) = i or 2 i + h
          const int n_idx = INL ? i * (N / A) + t * M + h * CPT + jj : (2 * i) * (N / A) + h * (N / A) + t * M + jj;
          const bool there = active && (p.in_valid == 0 || n_idx < p.in_valid);
          v[idx] = there ? row[jj] : c32{0.f, 0.f};
        } else {
          v[idx] = active ? row[jj] : c32{0.f, 0.f};
        }
      });
      }
    } else {
      static_for<0, NRAW>([&](auto ic) { raw[decltype(ic)::value] ^= xm_v; });   // int8 -> offset binary
    }
    // One transform (DC, window, three radix passes: v[] in sample order -> v[] in bin order) - or, CHIRP == 3, the whole
    // chirp-z convolution of a frame without leaving the workgroup: transform, times the chirp filter's spectrum,
    // conjugate, back through LDS into sample order, transform again (the complex64 intermediate of the two-launch version -
    // 16 bytes per point written and read - never reaches memory).
    if constexpr (CHIRP != 3 && CHIRP != 4) {
#include "tdsa_spectrum_passes.inc"
    } else {
      {
#include "tdsa_spectrum_passes.inc"
      }
    {
      // conj(X B) in natural bin order into LDS (element i at i + (i >> 5)), then the second transform's pass-1 inputs
      // (CHIRP == 4, rows of a long chirp-z frame: the filter spectrum has one row of N per k1 = frame mod out_mul_rows)
      const rsrc_t br = make_rsrc(p.out_mul + (CHIRP == 4 && p.out_mul_rows > 1 ? (long long)(frame % p.out_mul_rows) * N : 0ll), N * 8u);
      TDSA_SYNC();                                          // every thread has gathered its last-pass inputs
      static_for<0, 16>([&](auto ic) {
        constexpr int q = decltype(ic)::value;
        constexpr int kc = (q < 8 ? q : q + 8);
        const int k = t + 8 * h * SG + kc * SG;
        const u32x2 bq = __builtin_amdgcn_raw_buffer_load_b64(br, (unsigned(t) + 8u * unsigned(h) * SG) * 8u, kc * SG * 8u, 0);
        const c32 w = cmul(v[bitrev(q, 4)], c32{__uint_as_float(bq.x), __uint_as_float(bq.y)});
        lds_st(&buf[k + (k >> 5)], c32{w.x, -w.y});
      });
      TDSA_SYNC();
      static_for<0, 16>([&](auto ic) {
        constexpr int idx = decltype(ic)::value;
        constexpr int jj = idx / R1, i = idx % R1;
        const int n_idx = INL ? i * (N / A) + t * M + h * CPT + jj : (2 * i) * (N / A) + h * (N / A) + t * M + jj;
        v[idx] = lds_ld(&buf[n_idx + (n_idx >> 5)]);
      });
    }
      {
#include "tdsa_spectrum_passes.inc"
      }
    }
    TDSA_PRIO(1);
    TDSA_STAMP(10);

    // ---- epilogue: |X|^2 -> dB -> hold ; bin k = t + kc*SG lands at k ^ N/2 (fftshift) -----------
    // this thread's 16 bins: q < 8: kc = q + 8h (register bitrev(q)), q >= 8: kc = q + 8 + 8h (bitrev(q))
    if constexpr (CHIRP == 2 || CHIRP == 3) {
      static_assert(IN_C64 && HOLD == 0 && UNI, "fused chirp transforms: complex64 instantiation without hold, whole waves per frame");
      if (active) {
        // chirp-z plans, second transform: X[k] = a[k] conj(.) / M with |a[k]| = 1 - only |X|^2 of the bins k < post_n is
        // wanted: power / dB rows of post_n values, fftshift-ed as np.fft.fftshift does for any N, hold traces through
        // integer-punned atomics issued only where a trace moves (tdsa_chirp.hip step 4 folded into these stores; the
        // frame is wave-uniform: one SGPR descriptor per row, stores past its end - bins k >= post_n - are dropped)
        if constexpr (!C::WIN_LDS) load_window();
        const int pn = p.post_n, half = pn / 2;
        // bin k = t + 8 h SG + kc SG of this thread -> byte offset of its fftshift-ed position, or past the row's end
        auto bin_off = [&](int kc) -> unsigned {
          const int k = t + 8 * h * SG + kc * SG;
          int j = k + half;
          if (j >= pn) j -= pn;
          return k < pn ? unsigned(j) * 4u : 0xfffffff0u;
        };
        const float inv_m2 = p.post_inv_m * p.post_inv_m;
        if (p.out_lin != nullptr) {
          const rsrc_t lr = make_rsrc(p.out_lin + (long long)frame * pn, unsigned(pn) * 4u);
          const float sc = inv_m2 * p.pscale;
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            constexpr int kc = (q < 8 ? q : q + 8);
            const c32 X = v[bitrev(q, 4)];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((X.x * X.x + X.y * X.y) * sc), lr, bin_off(kc), 0, 0);
          });
        } else {
          // dB rows (the hold traces of a chirp-z plan are folded from these rows by chirp_hold_kernel: atomics in this
          // epilogue cost the instantiation 70 - 100 spilled registers)
          const bool mag = p.db_mode == 0;
          const rsrc_t orr = make_rsrc(p.out_db + (long long)frame * pn, unsigned(pn) * 4u);
          float db[16];
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            const c32 X = v[bitrev(q, 4)];
            db[q] = (X.x * X.x + X.y * X.y) * inv_m2;
          });
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            if (mag) db[q] = fmaf(2.0f * k10Log10_2, __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(db[q]) + p.log_floor), p.cal_db);
            else db[q] = fmaf(k10Log10_2, __builtin_amdgcn_logf(fmaf(db[q], p.pscale, p.log_floor)), p.cal_db);
          });
          if (p.tare != nullptr) {
            const rsrc_t tr = make_rsrc(p.tare, unsigned(pn) * 4u);
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              constexpr int kc = (q < 8 ? q : q + 8);
              db[q] -= __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tr, bin_off(kc), 0, 0));
            });
          }
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            constexpr int kc = (q < 8 ? q : q + 8);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db[q]), orr, bin_off(kc), 0, 0);
          });
        }
      }
    }
    else if constexpr (CHIRP == 1) {
      // first transform of a chirp-z plan: the spectrum leaves multiplied by the chirp filter's spectrum and conjugated,
      // ready for the inverse transform
      if (active) {
        if constexpr (!C::WIN_LDS) load_window();
        c32* crow = p.out_cplx + out_elem_off(frame) + t + 8 * h * SG;
        const c32* brow = p.out_mul + t + 8 * h * SG;
        static_for<0, 16>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          constexpr int kc = (q < 8 ? q : q + 8);
          const c32 z = cmul(v[bitrev(q, 4)], brow[kc * SG]);
          crow[kc * SG] = c32{z.x, -z.y};
        });
      } else {
        if constexpr (!C::WIN_LDS) load_window();
      }
    }
    else
    if (active) {
      if (p.out_cplx != nullptr) {            // real-input path: hand the complex bins to the fold kernel
        if constexpr (!C::WIN_LDS) load_window();
        c32* crow = p.out_cplx + out_elem_off(frame) + t + 8 * h * SG;
        bool plain = true;
        if constexpr (IN_C64 && HOLD == 0 && CHIRP != 4) {
          // chirp-z plans (tdsa_chirp.hip): the spectrum leaves multiplied by the chirp filter's spectrum and
          // conjugated, ready for the inverse transform - one pass over the rows less
          if (p.out_mul != nullptr) {
            plain = false;
            const c32* brow = p.out_mul + t + 8 * h * SG;
            if constexpr (FPW == 1) { if (p.out_mul_rows > 1) brow += (long long)(frame % p.out_mul_rows) * N; }
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              constexpr int kc = (q < 8 ? q : q + 8);
              const c32 z = cmul(v[bitrev(q, 4)], brow[kc * SG]);
              crow[kc * SG] = c32{z.x, -z.y};
            });
          } else if (p.out_valid != 0) {      // only the first out_valid bins are wanted (chirp-z: k < nfft of M)
            plain = false;
            const int k0 = t + 8 * h * SG;
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              constexpr int kc = (q < 8 ? q : q + 8);
              if (k0 + kc * SG < p.out_valid) crow[kc * SG] = v[bitrev(q, 4)];
            });
          }
        }
        if (plain) {
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            constexpr int kc = (q < 8 ? q : q + 8);             // + 8h folded into crow; natural order
            crow[kc * SG] = v[bitrev(q, 4)];
          });
        }
      } else if (p.out_lin != nullptr) {
        if constexpr (!C::WIN_LDS && !AGG) load_window();
        float* orow = p.out_lin + out_elem_off(frame) + t + 8 * h * SG;
        if constexpr (AGG) {
          const float w_f = in_vgpr(p.agg_w[frame]);          // FPW == 1: frame is workgroup-uniform, a scalar load
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
            const c32 X = v[bitrev(q, 4)];
            const float pw = (X.x * X.x + X.y * X.y) * ps_v;
            if (!p.agg_only) orow[kcs * SG] = pw;
            agg[q] = fmaf(w_f, pw, agg[q]);
          });
          if constexpr (!C::WIN_LDS) load_window();
        } else
        static_for<0, 16>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;       // shifted position of kc (without the 8h part)
          const c32 X = v[bitrev(q, 4)];
          orow[kcs * SG] = (X.x * X.x + X.y * X.y) * ps_v;
        });
      } else {
        // from here on only |X|^2 is needed: the 32 registers of v die as pw[] is formed
        float db[16];
        bool tiny = false;
        static_for<0, 16>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          const c32 X = v[bitrev(q, 4)];
          db[q] = X.x * X.x + X.y * X.y;
          tiny |= db[q] < kMagExactBelow;
        });
        if constexpr (STATS) {
          st_band = 0.f;
          st_maskc = st_mask;
          asm volatile("" : "+s"(st_maskc));
        }
        if constexpr (STATS) {
          // the band sum rides the dB loops only in waves that have bins in the band (a wave outside it took 32 branches per frame
          // for nothing).  (The loops of the other instantiations stay spelled out: wrapped in a lambda they moved the instruction
          // streams of the frozen ones.)
          unsigned st_partc = st_part;
          asm volatile("" : "+s"(st_partc));
          auto band_loops = [&](auto add) {
            if (mag_mode && __builtin_amdgcn_ballot_w64(tiny) != 0) {
              static_for<0, 16>([&](auto ic) {
                constexpr int q = decltype(ic)::value;
                const float mag = __builtin_amdgcn_sqrtf(db[q]);
                add(ic, (mag + p.log_floor) * (mag + p.log_floor));
                db[q] = fmaf(2.0f * k10Log10_2, __builtin_amdgcn_logf(mag + p.log_floor), cal_v);
              });
            } else if (mag_mode) {
              static_for<0, 16>([&](auto ic) {
                constexpr int q = decltype(ic)::value;
                add(ic, db[q]);
                db[q] = fmaf(k10Log10_2, __builtin_amdgcn_logf(db[q]), cal_v);
              });
            } else {
              static_for<0, 16>([&](auto ic) {
                constexpr int q = decltype(ic)::value;
                const float a = fmaf(db[q], ps_v, fl_v);
                add(ic, a);
                db[q] = fmaf(k10Log10_2, __builtin_amdgcn_logf(a), cal_v);
              });
            }
          };
          if ((st_maskc | st_partc) != 0u) {
            if (st_partc == 0u) band_loops(st_band_add);
            else band_loops(st_band_add_exact);
          } else {
          if (mag_mode && __builtin_amdgcn_ballot_w64(tiny) != 0) {   // near-silent frame: exact DB_MAG
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              const float mag = __builtin_amdgcn_sqrtf(db[q]);
              db[q] = fmaf(2.0f * k10Log10_2, __builtin_amdgcn_logf(mag + p.log_floor), cal_v);
            });
          } else if (mag_mode) {          // 10*log10(|X|^2): no power scale, no floor (wave-uniform branch)
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              db[q] = fmaf(k10Log10_2, __builtin_amdgcn_logf(db[q]), cal_v);
            });
          } else {
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              db[q] = fmaf(k10Log10_2, __builtin_amdgcn_logf(fmaf(db[q], ps_v, fl_v)), cal_v);
            });
          }
          }
        } else {
        if (mag_mode && __builtin_amdgcn_ballot_w64(tiny) != 0) {   // near-silent frame: exact DB_MAG
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            const float mag = __builtin_amdgcn_sqrtf(db[q]);
            db[q] = fmaf(2.0f * k10Log10_2, __builtin_amdgcn_logf(mag + p.log_floor), cal_v);
          });
        } else if (mag_mode) {          // 10*log10(|X|^2): no power scale, no floor (wave-uniform branch)
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            db[q] = fmaf(k10Log10_2, __builtin_amdgcn_logf(db[q]), cal_v);
          });
        } else {
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            db[q] = fmaf(k10Log10_2, __builtin_amdgcn_logf(fmaf(db[q], ps_v, fl_v)), cal_v);
          });
        }
        }
        if (p.tare != nullptr) {
          // SGPR descriptor + the store offsets: no per-thread 64-bit pointers (hoisted out of the frame loop
          // they cost 19 dwords of scratch per lane - 20 MB of spill writes per C3 launch)
          const rsrc_t tr = make_rsrc(p.tare, N * 4u);
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
            db[q] -= __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tr, out_voff, kcs * SG * 4u, 0));
          });
        }
        if constexpr (!C::WIN_LDS) load_window();     // next frame's window, ahead of this frame's stores
        TDSA_PRIO(0);
        if (p.out_db != nullptr) {
          float* orow = p.out_db + out_elem_off(frame);
          if constexpr (UNI) {
            const rsrc_t r = make_rsrc(orow, N * 4u);
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db[q]), r, out_voff, kcs * SG * 4u, kRowStorePolicy);
            });
          } else {
            float* orow_t = orow + t + 8 * h * SG;
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
              __builtin_nontemporal_store(db[q], &orow_t[kcs * SG]);
            });
          }
        }
        if constexpr ((HOLD & 3) != 0) {
          const bool nanfix = IN_C64 && (p.first_frame_index + frame == 0);
          static_for<0, 16>([&](auto ic) {
            constexpr int q = decltype(ic)::value;
            float dmx = db[q], dmn = db[q];
            if (nanfix && dmx != dmx) { dmx = -500.f; dmn = 500.f; }            // _nan_safe, first frame
            if constexpr ((HOLD & 1) != 0) hmax[q] = hw_max(hmax[q], dmx);      // np.fmax: NaN ignored
            if constexpr ((HOLD & 2) != 0) hmin[q] = hw_min(hmin[q], dmn);
          });
        }
        // (last in the epilogue: the dB values have no other reader left, the registers around them are at their fewest)
        if constexpr (STATS) {
          // the wave's record: {maximum of its 16 x 64 dB values, the first display position of the wave's bins that holds it
          // (np.argmax: first of equals; 0x40000000 | position of the first NaN - only complex64 input can carry one - and the
          // maximum is then NaN), the band's linear sum, 0}.  ONE lane holds the maximum in all but degenerate frames: its
          // sixteen bins are tested against it with one compare each, the holder's bit of every mask picked with
          // s_bitcmp.  Several holders (silence, flat spectra) or a NaN: one ballot per bin over the whole wave.
          // (That search for every wave and frame cost 17 us per C3 step; the holder's sixteen values stored for a later
          //  look-up - sixteen one-lane stores - 13; this: see profiles/r06_frame_stats.txt.)
          const float m = hw_max3(hw_max3(hw_max3(db[0], db[1], db[2]), hw_max3(db[3], db[4], db[5]), hw_max3(db[6], db[7], db[8])),
                                  hw_max3(hw_max3(db[9], db[10], db[11]), hw_max3(db[12], db[13], db[14]), db[15]), db[15]);
          const float mw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dpp_wave_max63(m)), 63));
          const unsigned long long holders = __builtin_amdgcn_ballot_w64(m == mw);
          const unsigned who = unsigned(__builtin_ctzll(holders));
          bool nan = false;
          if constexpr (IN_C64) {
            static_for<0, 16>([&](auto ic) { nan |= db[decltype(ic)::value] != db[decltype(ic)::value]; });
            nan = __builtin_amdgcn_ballot_w64(nan) != 0ull;
          }
          int best = 0x3fffffff;
          if (__builtin_popcountll(holders) == 1 && !nan) {      // wave-uniform
            // the holder parks its sixteen values in the wave's 64 bytes of LDS, in display order (rank r = kcs for
            // kcs < 8, kcs - 8 above), lanes 0 .. 15 read one each and compare: the first match is np.argmax's bin.
            // (Sixteen compares of all lanes with the holder's bit picked from each mask: 64 instructions; this: ~30.
            //  A wave's DS operations execute in order: the read needs no wait for the writes.)
            typedef __attribute__((address_space(3))) float* lds_fp;
            int wv = __builtin_amdgcn_readfirstlane(wave);
            asm volatile("" : "+s"(wv));
            // frames of several waves: 64 bytes per wave behind everything else; a frame inside one wave (N = 1024, where
            // four workgroups fill a CU's LDS to within 768 bytes): the head of the wave's own frame buffer, dead by now
            const unsigned st_addr = C::TPF <= 64 ? unsigned(uintptr_t((lds_fp)(buf)))
                                                  : unsigned(uintptr_t((lds_fp)(smem + C::LDS_ALLOC))) + 64u * unsigned(wv);
            unsigned long long ex_save;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %18\n\t"
                         "ds_write_b32 %17, %9 offset:0\n\tds_write_b32 %17, %10 offset:4\n\tds_write_b32 %17, %11 offset:8\n\t"
                         "ds_write_b32 %17, %12 offset:12\n\tds_write_b32 %17, %13 offset:16\n\tds_write_b32 %17, %14 offset:20\n\t"
                         "ds_write_b32 %17, %15 offset:24\n\tds_write_b32 %17, %16 offset:28\n\t"
                         "ds_write_b32 %17, %1 offset:32\n\tds_write_b32 %17, %2 offset:36\n\tds_write_b32 %17, %3 offset:40\n\t"
                         "ds_write_b32 %17, %4 offset:44\n\tds_write_b32 %17, %5 offset:48\n\tds_write_b32 %17, %6 offset:52\n\t"
                         "ds_write_b32 %17, %7 offset:56\n\tds_write_b32 %17, %8 offset:60\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(ex_save)
                         : "v"(db[0]), "v"(db[1]), "v"(db[2]), "v"(db[3]), "v"(db[4]), "v"(db[5]), "v"(db[6]), "v"(db[7]),
                           "v"(db[8]), "v"(db[9]), "v"(db[10]), "v"(db[11]), "v"(db[12]), "v"(db[13]), "v"(db[14]), "v"(db[15]),
                           "v"(st_addr), "s"(holders)
                         : "memory");
            float mine;                                   // (a ds_read of the wave's own slot: through a generic pointer it
            {                                             //  became a flat load behind vmcnt(0), its address a spilled pair)
              int tl = tid;
              asm volatile("" : "+v"(tl));
              const unsigned rd_addr = st_addr + unsigned(tl & 15) * 4u;
              asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(mine) : "v"(rd_addr) : "memory");
            }
            const unsigned hit = unsigned(__builtin_amdgcn_ballot_w64(mine == mw)) & 0xffffu;   // (never empty: the maximum is among them)
            const int r = __builtin_ctz(hit | 0x10000u);
            const int bk = r < 8 ? r : r + 8;
            best = (bk + 8 * int(who >> 5)) * SG + int(who & 31u);
          } else {
            asm volatile("; several holders / NaN" ::: "memory");    // (a side effect: this path must stay a branch, folded into
                                                                      //  selects it runs - sixteen more ballots - for every frame)
            static_for<0, 16>([&](auto ic) {
              constexpr int q = decltype(ic)::value;
              constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
              __builtin_amdgcn_sched_barrier(0);
              // (`>=`, not `==`: the same lanes - mw is the maximum - but not the same expression as the other path's, or all
              //  sixteen compares are hoisted above the branch as common subexpressions, into 32 SGPRs the kernel has not got)
              const unsigned long long mk = __builtin_amdgcn_ballot_w64(nan ? db[q] != db[q] : db[q] >= mw);
              if (mk != 0ull) {
                const int l = __builtin_ctzll(mk);                // lanes ascend with the position inside a half, halves with 8 SG
                const int cand = (kcs + 8 * (l >> 5)) * SG + (l & 31);
                best = cand < best ? cand : best;
              }
            });
          }
          __builtin_amdgcn_sched_barrier(0);
          int wq = __builtin_amdgcn_readfirstlane(wave);         // (the scalar wave index rebuilt per frame: no SGPR held for it)
          asm volatile("" : "+s"(wq));
          wq &= C::WPF - 1;
          const float bw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dpp_wave_sum63(st_band)), 63));
          const unsigned soff = (unsigned(frame) * C::WPF + unsigned(wq)) * unsigned(kStatsRecBytes);
          const rsrc_t sr = make_rsrc(p.stats_part, unsigned(p.n_frames) * (C::WPF * unsigned(kStatsRecBytes)));
          const u32x4 rec = {__float_as_uint(mw), (unsigned(best) + unsigned(wq) * 32u) | (nan ? 0x40000000u : 0u), __float_as_uint(bw), 0u};
          // lane 0 stores the record: exec narrowed around the store (a lane test on the thread index is hoisted out of
          // the frame loop as a mask the kernel has no SGPR pair for)
          unsigned long long ex_save;
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tbuffer_store_dwordx4 %1, off, %2, %3\n\ts_mov_b64 exec, %0"
                       : "=&s"(ex_save) : "v"(rec), "s"(sr), "s"(soff) : "memory");
        }
      }
    }
    else {
      if constexpr (!C::WIN_LDS) load_window();
    }
    TDSA_STAMP(11);
  }

#ifdef TDSA_TIMELINE
  if (p.dbg != nullptr && blockIdx.x == 0 && (tid & 63) == 0)
    tl[(2 * 16 + wave) * 16 + 14] = __builtin_amdgcn_s_memtime();         // frame loop left
  __syncthreads();
  if (p.dbg != nullptr && blockIdx.x == 0)
    for (int i = tid; i < 8 * 16 * 16; i += C::WGT) p.dbg[i] = tl[i];
#endif
  // fold this workgroup's register-resident hold traces straight into the plan's traces with
  // integer-punned float atomics (max/min are associative; the traces start at -inf / +inf).
  // (Issuing the bulk of them before the last frame to hide the tail was tried: the extra live state
  //  pushed the kernel into 39 scratch spills and cost more than the ~6 us tail it removed.  Fetching the
  //  current trace into the window registers during the last frame, to save the round trip below: 16 bytes
  //  of scratch inside the frame loop, 75.3 instead of 73.9 us per C3 launch.)
  if constexpr (AGG && FPW == 1) {
    if (u1 > u0) {                          // (a workgroup without frames leaves no row: the chain skips empty chunks)
      int tid_a = threadIdx.x;
      asm volatile("" : "+v"(tid_a));
      const int h_a = (tid_a >> 5) & 1, t_a = (tid_a >> 6) * 32 + (tid_a & 31);
      float* arow = p.agg_out + (long long)blockIdx.x * N + t_a + 8 * h_a * SG;
      static_for<0, 16>([&](auto ic) {
        constexpr int q = decltype(ic)::value;
        constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
        arow[kcs * SG] = agg[q];
      });
    }
  } else if constexpr (AGG) {
    // the slots' partial sums (weights of the workgroup's whole range: they simply add) meet in LDS, in slot order
    float* lagg = reinterpret_cast<float*>(smem);                 // [FPW][N], display order
    __syncthreads();                                              // every slot is done with the frame buffers
    static_for<0, 16>([&](auto ic) {
      constexpr int q = decltype(ic)::value;
      constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
      lagg[slot * N + t + 8 * h * SG + kcs * SG] = agg[q];
    });
    __syncthreads();
    if (u1 > u0) {
      for (int i = tid; i < N; i += C::WGT) {
        float sum = lagg[i];
#pragma unroll
        for (int sl = 1; sl < FPW; ++sl) sum += lagg[sl * N + i];
        p.agg_out[(long long)blockIdx.x * N + i] = sum;
      }
    }
  }
  if constexpr ((HOLD & 3) != 0) {
    // the row index is rebuilt from a fresh (opaque) copy of the thread index: kept live across the frame loop
    // `8 * h * SG` was the one value of the C3 instantiation that did not fit the 128 VGPRs (one dword of
    // scratch per lane = 2.1 MB of spill writes per launch, WRITE_SIZE 162.1 instead of 160 MB)
    int tid_m = threadIdx.x;
    asm volatile("" : "+v"(tid_m));
    const int h_m = (tid_m >> 5) & 1;
    const int g_m = (tid_m >> 6) * 32 + (tid_m & 31);
    const int t_m = (FPW == 1) ? g_m : g_m - (g_m / SG) * SG;
    const int prow = t_m + 8 * h_m * SG;
    const bool any = (u1 > u0) && (FPW == 1 || u0 * FPW + slot < p.n_frames);
    if constexpr (FPW == 1) {
      if (any) {
        // The traces only ever grow (shrink), so a value read earlier - even a stale one from this XCD's
        // L2 - is a valid lower (upper) bound: the atomic is issued only when this workgroup would
        // actually move the trace.  In a long capture that is rare, and the 4.2 M device-scope atomics
        // per C3 launch (5-10 us) all but disappear.
        float cmax[(HOLD & 1) ? 16 : 1], cmin[(HOLD & 2) ? 16 : 1];
        static_for<0, 16>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
          if constexpr ((HOLD & 1) != 0) cmax[q] = p.part_max[prow + kcs * SG];
          if constexpr ((HOLD & 2) != 0) cmin[q] = p.part_min[prow + kcs * SG];
        });
        static_for<0, 16>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
          if constexpr ((HOLD & 1) != 0) {
            if (hmax[q] > cmax[q]) atomic_fmax_dev(p.part_max + prow + kcs * SG, hmax[q]);
          }
          if constexpr ((HOLD & 2) != 0) {
            if (hmin[q] < cmin[q]) atomic_fmin_dev(p.part_min + prow + kcs * SG, hmin[q]);
          }
        });
      }
    } else {
      // several frames per workgroup hold the same bins: fold them in LDS first, then one global atomic
      // per bin and workgroup (at N = 64 the direct version put 65536 atomics on every address)
      float* lmax = reinterpret_cast<float*>(smem);
      float* lmin = lmax + N;
      __syncthreads();
      for (int i = tid; i < N; i += C::WGT) {
        lmax[i] = -INFINITY;
        lmin[i] = INFINITY;
      }
      __syncthreads();
      if (any) {
        static_for<0, 16>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          constexpr int kcs = (q < 8 ? q : q + 8) ^ 16;
          if constexpr ((HOLD & 1) != 0) atomic_fmax_dev(lmax + prow + kcs * SG, hmax[q]);
          if constexpr ((HOLD & 2) != 0) atomic_fmin_dev(lmin + prow + kcs * SG, hmin[q]);
        });
      }
      __syncthreads();
      if (u1 > u0) {
        for (int i = tid; i < N; i += C::WGT) {
          if constexpr ((HOLD & 1) != 0) {
            if (lmax[i] > p.part_max[i]) atomic_fmax_dev(p.part_max + i, lmax[i]);
          }
          if constexpr ((HOLD & 2) != 0) {
            if (lmin[i] < p.part_min[i]) atomic_fmin_dev(p.part_min + i, lmin[i]);
          }
        }
      }
    }
  }
#ifdef TDSA_TIMELINE
  if (p.dbg != nullptr && blockIdx.x == 0 && (tid & 63) == 0)
    p.dbg[(2 * 16 + wave) * 16 + 15] = __builtin_amdgcn_s_memtime();      // hold traces merged: the wave ends here
#endif
}

// window table -> the order the frame kernel's threads consume it: value idx = c*R1 + r of thread (t, h) of a frame,
// w[a(r)*(N/A) + t*M + c (+ h*CPT)] with a(r) = r (INL) or 2r + h, sits at wp[((idx/4)*2SG + h*SG + t)*4 + idx%4]
template <int LOG2N>
__global__ void __launch_bounds__(256) window_perm_kernel(const float* w, float* wp) {
  using C = Cfg<LOG2N>;
  constexpr int N = C::N, SG = C::SG, A = C::A, M = C::M, R1 = C::R1, CPT = C::CPT;
  const int th = blockIdx.x * 256 + threadIdx.x;
  if (th >= 2 * SG) return;
  const int h = th / SG, t = th - h * SG;
#pragma unroll
  for (int idx = 0; idx < 16; ++idx) {
    const int c = idx / R1, r = idx % R1;
    const int a = C::INL ? r : 2 * r + h;
    wp[((idx >> 2) * 2 * SG + th) * 4 + (idx & 3)] = w[a * (N / A) + t * M + (C::INL ? h * CPT : 0) + c];
  }
}
template <int LOG2N>
inline hipError_t perm_for(const float* w, float* wp, hipStream_t s) {
  if constexpr (Cfg<LOG2N>::WIN_LDS) return hipSuccess;       // those sizes read the natural table into LDS
  hipLaunchKernelGGL(window_perm_kernel<LOG2N>, dim3((2 * Cfg<LOG2N>::SG + 255) / 256), dim3(256), 0, s, w, wp);
  return hipGetLastError();
}

template <int LOG2N>
inline LaunchGeom geom_for(int n_frames, int num_cu) {
  using C = Cfg<LOG2N>;
  LaunchGeom g;
  g.block = C::WGT;
  g.fpw = C::FPW;
  g.lds_bytes = C::LDS_ALLOC;
  int per_cu = int((160 * 1024) / C::LDS_BYTES);
  if (per_cu < 1) per_cu = 1;
  const int by_waves = 4 * 4 * 64 / C::WGT;                   // kernel is built for 4 waves per SIMD
  const int wg_per_cu = per_cu < by_waves ? per_cu : by_waves;
  const int units = (n_frames + C::FPW - 1) / C::FPW;
  int grid = num_cu * wg_per_cu;
  if (grid > units) grid = units;
  if (grid < 1) grid = 1;
  g.grid = grid;
  return g;
}

template <int LOG2N, bool IN_C64, int HOLD, int CHIRP = 0>
inline hipError_t launch_one(const SpecParams& p, const LaunchGeom& g, hipStream_t s) {
  auto k = spectrum_kernel<LOG2N, IN_C64, HOLD, CHIRP>;
  static std::atomic<unsigned long long> attr_done{0};
  // (STATS instantiations: 64 bytes per wave behind everything else - the sixteen dB values of the lane that holds the wave's maximum)
  const size_t lds = g.lds_bytes + ((HOLD & 8) != 0 && Cfg<LOG2N>::TPF > 64 ? size_t(Cfg<LOG2N>::NWAVE) * 64 : 0);
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), int(lds), attr_done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(g.grid), dim3(g.block), lds, s, p);
  return hipGetLastError();
}

template <int LOG2N>
hipError_t launch_n(int in_c64, const SpecParams& p, const LaunchGeom& g, hipStream_t s) {
  const int hold = p.out_lin == nullptr ? (p.hold_flags & 3) : 0;
  if (p.out_lin != nullptr && p.agg_out != nullptr)
    return in_c64 ? launch_one<LOG2N, true, 4>(p, g, s) : launch_one<LOG2N, false, 4>(p, g, s);
  if constexpr (Cfg<LOG2N>::TPF >= 64) {       // chirp-z plans: the transforms that carry the element-wise passes
    if constexpr (LOG2N == 14) {
      if (in_c64 && p.rows_twice != 0) return launch_one<LOG2N, true, 0, 4>(p, g, s);
    }
    if (in_c64 && p.pre_raw != nullptr && p.post_n != 0) return launch_one<LOG2N, true, 0, 3>(p, g, s);
#ifdef TDSA_DEV   // the two transforms as two launches, each carrying one element-wise pass: developer builds only (A/B)
    if (in_c64 && p.pre_raw != nullptr) return launch_one<LOG2N, true, 0, 1>(p, g, s);
    if (in_c64 && p.post_n != 0) return launch_one<LOG2N, true, 0, 2>(p, g, s);
#endif
  }
  if (p.pre_raw != nullptr || p.post_n != 0 || p.rows_twice != 0) return hipErrorInvalidValue;   // no such instantiation
  if (p.stats_part != nullptr) {          // per-frame scalars from the epilogue (the host asks only where these exist)
    if constexpr (Cfg<LOG2N>::TPF >= 64) {
      if (hold > 1 || p.tare != nullptr || p.out_cplx != nullptr || p.out_lin != nullptr) return hipErrorInvalidValue;
      if (in_c64) return hold ? launch_one<LOG2N, true, 9>(p, g, s) : launch_one<LOG2N, true, 8>(p, g, s);
      return hold ? launch_one<LOG2N, false, 9>(p, g, s) : launch_one<LOG2N, false, 8>(p, g, s);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (in_c64) {
    switch (hold) {
      case 0: return launch_one<LOG2N, true, 0>(p, g, s);
      case 1: return launch_one<LOG2N, true, 1>(p, g, s);
      case 2: return launch_one<LOG2N, true, 2>(p, g, s);
      default: return launch_one<LOG2N, true, 3>(p, g, s);
    }
  }
  switch (hold) {
    case 0: return launch_one<LOG2N, false, 0>(p, g, s);
    case 1: return launch_one<LOG2N, false, 1>(p, g, s);
    case 2: return launch_one<LOG2N, false, 2>(p, g, s);
    default: return launch_one<LOG2N, false, 3>(p, g, s);
  }
}

// one translation unit per size (tdsa_spectrum_inst.hip, -DTDSA_LOG2N=k) provides these
template <int LOG2N> hipError_t launch_size(int in_c64, const SpecParams& p, const LaunchGeom& g, hipStream_t s);
template <int LOG2N> LaunchGeom geom_size(int n_frames, int num_cu);
template <int LOG2N> hipError_t perm_size(const float* w, float* wp, hipStream_t s);

}  // namespace tdsa
