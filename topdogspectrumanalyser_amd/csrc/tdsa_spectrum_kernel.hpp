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
//   * SG = N/32 threads own one frame; each thread holds 32 complex points in VGPRs in every pass.
//   * pass 1: M = 32/A adjacent radix-A butterflies per thread, inputs straight from global memory
//             (thread t reads samples a*(N/A) + t*M .. +M-1 : 2M contiguous bytes per load), no twiddles.
//             The raw registers are refilled with the NEXT frame's samples as soon as they have been
//             unpacked, so the HBM read of frame f+1 overlaps the FFT of frame f at no register cost.
//   * pass 2 (3-pass sizes): radix-32 IN PLACE (a thread stores output kb into the LDS slot input b = kb
//             came from, so no barrier separates its gather from its scatter), twiddles W_(32A)^(ka*b)
//             read from a 4 KiB LDS table.  Outputs are stored depth-first, as soon as they are final,
//             so the slow LDS write path (~80 B/clk/CU) drains underneath the remaining butterflies.
//   * last pass: radix-32, twiddles W_N^(t*c) rebuilt from 10 exact per-thread table values
//             (every factor is at most one rounded product away from the table).
//   * the thread -> bin mapping of the last pass is frame invariant, so the window, the twiddle seeds
//     and the max/min hold traces live in registers across the persistent frame loop.
//   * a workgroup processes a CONTIGUOUS range of frames (overlapping frames re-read their shared
//     half from the same XCD's L2, not from HBM).
#pragma once
#include "tdsa_fft.hpp"
#include "tdsa_kernels.hpp"

namespace tdsa {

template <int LOG2N>
struct Cfg {
  static constexpr int N = 1 << LOG2N;
  static constexpr int SG = N / 32;                       // threads per frame
  static constexpr int NPASS = (N >= 2048) ? 3 : 2;
  static constexpr int A = (NPASS == 3) ? N / 1024 : N / 32;  // radix of the first pass
  static constexpr int M = 32 / A;                        // adjacent first-pass butterflies per thread
  static constexpr int WGT = SG > 256 ? SG : 256;         // threads per workgroup
  static constexpr int FPW = WGT / SG;                    // frames in flight per workgroup
  static constexpr int NPAD = N + N / 32;                 // LDS slot: rows of 32 complex + 1 pad element
  static constexpr int WPF = SG >= 64 ? SG / 64 : 1;      // waves per frame
  static constexpr int NWAVE = WGT / 64;
  static constexpr int TWM = (NPASS == 3) ? 32 * A : 0;   // middle-pass twiddle table entries [b][ka]
  static constexpr size_t DATA_BYTES = size_t(FPW) * NPAD * sizeof(c32);
  static constexpr size_t LDS_BYTES = DATA_BYTES + size_t(TWM) * sizeof(c32) + NWAVE * 2 * sizeof(double);
  static constexpr size_t LDS_ALLOC = LDS_BYTES
#ifdef TDSA_TIMELINE
      + 8 * 8 * 16 * 8
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

// wave64 integer sum with DPP adds (no LDS round trips); the total is valid in lanes 48..63
__device__ __forceinline__ int dpp_wave_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);   // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);   // row_mirror: every lane = row sum
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return x;
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

constexpr float k10Log10_2 = 3.01029995663981195214f;   // 10*log10(2)
constexpr float kMagExactBelow = 1e-8f;                  // |X|^2 below this: the 1e-12 floor is visible

// developer ablation builds (-DTDSA_ABLATE=mask): 1 no barriers, 2 no LDS exchange, 4 no dB stores,
// 8 no raw/window loads.  Results are wrong by construction; timing only.
#ifndef TDSA_ABLATE
#define TDSA_ABLATE 0
#endif
#define TDSA_SYNC()                                  \
  do {                                               \
    if constexpr ((TDSA_ABLATE & 1) == 0) __syncthreads(); \
  } while (0)

#ifdef TDSA_TIMELINE
// developer build: wave 0..7 of workgroup 0 stamp s_memtime into LDS at phase boundaries (first 8 frames)
#define TDSA_STAMP(i)                                                                              \
  do {                                                                                             \
    if (p.dbg != nullptr && blockIdx.x == 0 && (tid & 63) == 0 && unit - u0 < 8)                   \
      tl[((unit - u0) * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime();                      \
  } while (0)
#else
#define TDSA_STAMP(i)
#endif

template <int LOG2N, bool IN_C64, int HOLD>   // HOLD: bit0 = max trace, bit1 = min trace
__global__ void __launch_bounds__(Cfg<LOG2N>::WGT, 2) spectrum_kernel(const SpecParams p) {
  using C = Cfg<LOG2N>;
  constexpr int N = C::N, SG = C::SG, A = C::A, M = C::M, FPW = C::FPW, NPAD = C::NPAD;
  constexpr int LA = ilog2(A);
  constexpr int DW = (M >= 2) ? M / 2 : 1;   // raw dwords per first-pass row
  constexpr int NRAW = IN_C64 ? 1 : A * DW;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c32* lds = reinterpret_cast<c32*>(smem);
  c32* twm = reinterpret_cast<c32*>(smem + C::DATA_BYTES);
  double* red = reinterpret_cast<double*>(smem + C::DATA_BYTES + size_t(C::TWM) * sizeof(c32));
#ifdef TDSA_TIMELINE
  unsigned long long* tl = reinterpret_cast<unsigned long long*>(smem + C::LDS_BYTES);   // 8 KiB extra
#endif

  const int tid = threadIdx.x;
  const int slot = (FPW == 1) ? 0 : tid / SG;
  const int t = (FPW == 1) ? tid : tid - slot * SG;
  const int wave = tid >> 6;
  c32* buf = lds + slot * NPAD;

  const int n_units = (p.n_frames + FPW - 1) / FPW;
  const int u0 = int((long long)blockIdx.x * n_units / gridDim.x);
  const int u1 = int((long long)(blockIdx.x + 1) * n_units / gridDim.x);

  // ---- frame-invariant per-thread state ---------------------------------------------------------
  // the window is re-read from L2 every frame (64 KiB table shared by all workgroups): keeping its
  // 32 values per thread in VGPRs pushed the hold variants into scratch spills
  const rsrc_t win_rsrc = make_rsrc(p.window, N * 4u);
  const unsigned win_voff = unsigned(t) * (M * 4u);
  c32 twf_lo[3], twf_hi[7];   // last pass seeds: W_N^(t*c), c = 1,2,3 and 4,8,..,28
  static_for<0, 3>([&](auto ic) { constexpr int i = decltype(ic)::value; twf_lo[i] = p.tw[t * (i + 1)]; });
  static_for<0, 7>([&](auto ic) { constexpr int i = decltype(ic)::value; twf_hi[i] = p.tw[t * 4 * (i + 1)]; });
  if constexpr (C::NPASS == 3) {   // middle pass table twm[b*A + ka] = W_(32A)^(ka*b)
    if (tid < C::TWM) {
      const int b = tid / A, ka = tid % A;
      twm[tid] = p.tw[ka * b * (N / (32 * A))];
    }
  }
  float hmax[(HOLD & 1) ? 32 : 1], hmin[(HOLD & 2) ? 32 : 1];
  static_for<0, 32>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr ((HOLD & 1) != 0) hmax[i] = -INFINITY;
    if constexpr ((HOLD & 2) != 0) hmin[i] = INFINITY;
  });
  // epilogue constants in VGPRs.  DB_MAG is evaluated as 10*log10(|X|^2): identical to
  // 20*log10(|X| + 1e-12) in float32 whenever |X|^2 >= 1e-8 (the floor is below half an ulp of |X|);
  // frames with a smaller bin take the exact path below.
  const unsigned xm_v = [&] { unsigned x = (M == 1) ? (p.xor_mask & 0xffffu) : p.xor_mask; asm volatile("" : "+v"(x)); return x; }();
  const bool mag_mode = p.db_mode == 0;
  const float ps_v = in_vgpr(mag_mode ? 1.0f : p.pscale);
  const float fl_v = in_vgpr(mag_mode ? 0.0f : p.log_floor);
  const float cal_v = in_vgpr(p.cal_db);

  // LDS addressing in complex elements: element i of a pass lives at i + (i >> 5)
  const int wr1_base = 33 * t;                                   // pass 1 writes row t
  int rd_base = 0;                                               // gather of y[t + b*SG]
  constexpr int rd_stride = SG + SG / 32;
  if constexpr (SG % 32 == 0) rd_base = t + (t >> 5);
  const int ka_mid = t % A;
  const int rd3_base = (t / A) * rd_stride + ka_mid;             // last gather after the in-place pass

  const unsigned lane_in_off = unsigned(t) * (IN_C64 ? M * 8u : M * 2u);   // byte offset inside a row

  uint32_t raw[NRAW];
  auto load_frame_raw = [&](int frame) {
    if constexpr (!IN_C64) {
      const bool act = frame < p.n_frames;
      const unsigned char* fb = static_cast<const unsigned char*>(p.in) + (long long)frame * p.frame_stride;
      if constexpr ((TDSA_ABLATE & 8) != 0) {
        static_for<0, NRAW>([&](auto ic) { raw[decltype(ic)::value] = 0x01020304u * (t + 1); });
      } else if constexpr (FPW == 1 && M >= 2) {   // frame is workgroup-uniform: SGPR descriptor + lane offset
        const rsrc_t r = make_rsrc(fb, N * 2u);
        static_for<0, A>([&](auto ic) {
          constexpr int a = decltype(ic)::value;
          buf_load<DW>(r, lane_in_off, a * (N / A) * 2u, &raw[a * DW]);
        });
      } else {
        static_for<0, A>([&](auto ic) {
          constexpr int a = decltype(ic)::value;
          const unsigned char* row = fb + a * (N / A) * 2;
          if (act) {
            if constexpr (M == 1) raw[a] = *reinterpret_cast<const uint16_t*>(row + lane_in_off);
            else load_raw<DW>(row + lane_in_off, &raw[a * DW]);
          } else {
            static_for<0, DW>([&](auto jc) { raw[a * DW + decltype(jc)::value] = 0u; });
          }
        });
      }
    }
  };
  if (u0 < u1) load_frame_raw(u0 * FPW + slot);

  for (int unit = u0; unit < u1; ++unit) {
    const int frame = unit * FPW + slot;
    const bool active = frame < p.n_frames;

    TDSA_STAMP(0);
    c32 v[32];
    float sub_re = p.in_off, sub_im = p.in_off;
    float win[32];                                          // win[jj*A + a] = w[a*(N/A) + t*M + jj]
    static_for<0, A>([&](auto ic) {
      constexpr int a = decltype(ic)::value;
      uint32_t wq[M];
      if constexpr ((TDSA_ABLATE & 8) != 0) { static_for<0, M>([&](auto jc) { wq[decltype(jc)::value] = 0x3f800000u; }); }
      else buf_load<M>(win_rsrc, win_voff, a * (N / A) * 4u, wq);
      static_for<0, M>([&](auto jc) { constexpr int jj = decltype(jc)::value; win[jj * A + a] = __uint_as_float(wq[jj]); });
    });

    // ---- frame sums for DC removal ---------------------------------------------------------------
    if constexpr (IN_C64) {
      const unsigned char* fb = static_cast<const unsigned char*>(p.in) + (long long)frame * p.frame_stride;
      static_for<0, 32>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int jj = i / A, a = i % A;
        const c32* row = reinterpret_cast<const c32*>(fb + a * (N / A) * 8 + lane_in_off);
        v[i] = active ? row[jj] : c32{0.f, 0.f};
      });
    } else {
      static_for<0, NRAW>([&](auto ic) { raw[decltype(ic)::value] ^= xm_v; });   // int8 -> offset binary
    }
    if constexpr (!IN_C64 && SG >= 64) {
      // byte formats, whole waves per frame: exact integer sums, DPP wave reduce, one int2 per wave
      int* redi = reinterpret_cast<int*>(red);
      if (p.dc_mode == DC_FRAME_MEAN) {
        unsigned si = 0, sq = 0;
        static_for<0, NRAW>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          si = __builtin_amdgcn_udot4(raw[i], 0x00010001u, si, false);
          sq = __builtin_amdgcn_udot4(raw[i], 0x01000100u, sq, false);
        });
        const int wi = dpp_wave_sum(int(si)), wq = dpp_wave_sum(int(sq));
        if ((tid & 63) == 63) *reinterpret_cast<int2*>(&redi[wave * 2]) = int2{wi, wq};
      }
      TDSA_STAMP(1);
      TDSA_SYNC();       // also the WAR fence between the previous frame's LDS reads and our writes
      TDSA_STAMP(2);
      if (p.dc_mode == DC_FRAME_MEAN) {
        const int w0 = (slot * SG) >> 6;
        int part[2 * C::WPF];
        if constexpr (C::WPF % 2 == 0) {
          static_for<0, C::WPF / 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int4 q = *reinterpret_cast<const int4*>(&redi[(w0 + 2 * i) * 2]);
            part[4 * i] = q.x; part[4 * i + 1] = q.y; part[4 * i + 2] = q.z; part[4 * i + 3] = q.w;
          });
        } else {
          const int2 q = *reinterpret_cast<const int2*>(&redi[w0 * 2]);
          part[0] = q.x; part[1] = q.y;
        }
        int ti = 0, tq = 0;
        static_for<0, C::WPF>([&](auto ic) { constexpr int i = decltype(ic)::value; ti += part[2 * i]; tq += part[2 * i + 1]; });
        sub_re = float(ti) * (1.0f / N);     // exact: sums < 2^24, N a power of two
        sub_im = float(tq) * (1.0f / N);
        if (p.dc_state != nullptr && frame == p.n_frames - 1 && t == 0)
          *p.dc_state = c32{(sub_re - p.in_off) * p.in_scale, (sub_im - p.in_off) * p.in_scale};
      } else if (p.dc_mode == DC_TRACKED && active) {
        const c32 sv = p.dc_sub[frame]; sub_re = sv.x; sub_im = sv.y;
      }
    } else if (p.dc_mode == DC_FRAME_MEAN) {
      constexpr int W = SG < 64 ? SG : 64;
      double s_re, s_im;
      if constexpr (IN_C64) {
        s_re = 0.0; s_im = 0.0;
        static_for<0, 32>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          s_re += double(v[i].x); s_im += double(v[i].y);
        });
        s_re = seg_sum<W>(s_re); s_im = seg_sum<W>(s_im);
      } else {
        unsigned si = 0, sq = 0;
        static_for<0, NRAW>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          si = __builtin_amdgcn_udot4(raw[i], 0x00010001u, si, false);
          sq = __builtin_amdgcn_udot4(raw[i], 0x01000100u, sq, false);
        });
        s_re = double(seg_sum<W>(int(si))); s_im = double(seg_sum<W>(int(sq)));
      }
      if constexpr (SG > 64) {
        if ((tid & 63) == 0) { red[wave * 2] = s_re; red[wave * 2 + 1] = s_im; }
      }
      __syncthreads();   // also the WAR fence between the previous frame's LDS reads and our writes
      if constexpr (SG > 64) {
        const int w0 = (slot * SG) >> 6;
        s_re = 0.0; s_im = 0.0;
#pragma unroll
        for (int i = 0; i < C::WPF; ++i) { s_re += red[(w0 + i) * 2]; s_im += red[(w0 + i) * 2 + 1]; }
      }
      sub_re = float(s_re * (1.0 / N));
      sub_im = float(s_im * (1.0 / N));
      if (p.dc_state != nullptr && frame == p.n_frames - 1 && t == 0)
        *p.dc_state = c32{(sub_re - p.in_off) * p.in_scale, (sub_im - p.in_off) * p.in_scale};
    } else {
      __syncthreads();
      if (p.dc_mode == DC_TRACKED && active) { const c32 s = p.dc_sub[frame]; sub_re = s.x; sub_im = s.y; }
    }
    sub_re = in_vgpr(sub_re);
    sub_im = in_vgpr(sub_im);

    // ---- unpack + DC removal + window ------------------------------------------------------------
    if constexpr (IN_C64) {
      static_for<0, 32>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        v[i] = c32{(v[i].x - sub_re) * win[i], (v[i].y - sub_im) * win[i]};
      });
    } else if constexpr (M == 1) {
      static_for<0, 32>([&](auto ic) {
        constexpr int a = decltype(ic)::value;
        const uint32_t u = raw[a];
        v[a] = c32{(float(u & 0xffu) - sub_re) * win[a], (float((u >> 8) & 0xffu) - sub_im) * win[a]};
      });
    } else {
      static_for<0, A>([&](auto ac) {
        constexpr int a = decltype(ac)::value;
        static_for<0, DW>([&](auto dc) {
          constexpr int d = decltype(dc)::value;
          const uint32_t u = raw[a * DW + d];
          constexpr int i0 = (2 * d) * A + a, i1 = (2 * d + 1) * A + a;
          v[i0] = c32{(float(u & 0xffu) - sub_re) * win[i0], (float((u >> 8) & 0xffu) - sub_im) * win[i0]};
          v[i1] = c32{(float((u >> 16) & 0xffu) - sub_re) * win[i1], (float(u >> 24) - sub_im) * win[i1]};
        });
      });
    }
    TDSA_STAMP(3);
    // raw registers are free again: start the next frame's HBM read now, it lands during the FFT
    if (unit + 1 < u1) load_frame_raw((unit + 1) * FPW + slot);

    // ---- pass 1: M radix-A butterflies; each output is stored the moment it is final ------------
    static_for<0, M>([&](auto jc) {
      constexpr int jj = decltype(jc)::value;
      dif_emit<A, jj * A, 32>(v, [&](auto rc) {
        constexpr int r = decltype(rc)::value;                 // register jj*A + bitrev(ka)
        constexpr int ka = bitrev(r - jj * A, LA);
        if constexpr ((TDSA_ABLATE & 2) == 0) buf[wr1_base + jj * A + ka] = v[r];
      });
    });
    TDSA_STAMP(4);
    TDSA_SYNC();
    TDSA_STAMP(5);

    // ---- middle radix-32 pass (3-pass sizes), IN PLACE: thread t owns the 32 slots it gathers -----
    if constexpr (C::NPASS == 3) {
      static_for<0, 32>([&](auto ic) {
        constexpr int b = decltype(ic)::value;
        if constexpr ((TDSA_ABLATE & 2) == 0) v[b] = buf[rd_base + b * rd_stride];
      });
      TDSA_STAMP(6);
      int ka_o = ka_mid;
      asm volatile("" : "+v"(ka_o));                        // keep the table reads inside the loop
      static_for<1, 32>([&](auto ic) {
        constexpr int b = decltype(ic)::value;
        v[b] = cmul(v[b], twm[b * A + ka_o]);
      });
      dif_emit<32, 0, 32>(v, [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int kb = bitrev(r, 5);
        if constexpr ((TDSA_ABLATE & 2) == 0) buf[rd_base + kb * rd_stride] = v[r];   // slot of element b = kb
      });
      TDSA_STAMP(7);
      TDSA_SYNC();
      TDSA_STAMP(8);
      static_for<0, 32>([&](auto ic) {                         // element (c, kb, ka) for this (kb, ka)
        constexpr int c = decltype(ic)::value;
        if constexpr ((TDSA_ABLATE & 2) == 0) v[c] = buf[rd3_base + c * A + ((c * A) >> 5)];
      });
    } else {
      static_for<0, 32>([&](auto ic) {
        constexpr int b = decltype(ic)::value;
        if constexpr (SG % 32 == 0) v[b] = buf[rd_base + b * rd_stride];
        else { const int i = t + b * SG; v[b] = buf[i + (i >> 5)]; }
      });
    }
    TDSA_STAMP(9);
    static_for<0, 3>([&](auto ic) { opaque(twf_lo[decltype(ic)::value]); });
    static_for<0, 7>([&](auto ic) { opaque(twf_hi[decltype(ic)::value]); });
    twiddle32(v, twf_lo, twf_hi);
    dif<32, 0, 32>(v);
    TDSA_STAMP(10);

    // ---- epilogue: |X|^2 -> dB -> hold ; bin k = t + kc*SG lands at k ^ N/2 (fftshift) -----------
    if (active) {
      if (p.out_lin != nullptr) {
        float* orow = p.out_lin + (long long)frame * N;
        static_for<0, 32>([&](auto ic) {
          constexpr int kc = decltype(ic)::value;
          const c32 X = v[bitrev(kc, 5)];
          (orow + (kc ^ 16) * SG)[t] = (X.x * X.x + X.y * X.y) * ps_v;
        });
      } else {
        float db[32];
        bool tiny = false;
        static_for<0, 32>([&](auto ic) {
          constexpr int kc = decltype(ic)::value;
          const c32 X = v[bitrev(kc, 5)];
          const float pw = X.x * X.x + X.y * X.y;
          tiny |= pw < kMagExactBelow;
          db[kc] = fmaf(k10Log10_2, __builtin_amdgcn_logf(fmaf(pw, ps_v, fl_v)), cal_v);
        });
        if (mag_mode && __builtin_amdgcn_ballot_w64(tiny) != 0) {   // near-silent frame: exact DB_MAG
          static_for<0, 32>([&](auto ic) {
            constexpr int kc = decltype(ic)::value;
            const c32 X = v[bitrev(kc, 5)];
            const float mag = __builtin_amdgcn_sqrtf(X.x * X.x + X.y * X.y);
            db[kc] = fmaf(2.0f * k10Log10_2, __builtin_amdgcn_logf(mag + p.log_floor), cal_v);
          });
        }
        if (p.tare != nullptr) {
          static_for<0, 32>([&](auto ic) {
            constexpr int kc = decltype(ic)::value;
            db[kc] -= (p.tare + (kc ^ 16) * SG)[t];
          });
        }
        if ((TDSA_ABLATE & 4) == 0 && p.out_db != nullptr) {
          float* orow = p.out_db + (long long)frame * N;
          if constexpr (FPW == 1) {
            const rsrc_t r = make_rsrc(orow, N * 4u);
            static_for<0, 32>([&](auto ic) {
              constexpr int kc = decltype(ic)::value;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db[kc]), r, unsigned(t) * 4u,
                                                    (kc ^ 16) * SG * 4u, 0);
            });
          } else {
            static_for<0, 32>([&](auto ic) {
              constexpr int kc = decltype(ic)::value;
              (orow + (kc ^ 16) * SG)[t] = db[kc];
            });
          }
        }
        if constexpr (HOLD != 0) {
          const bool nanfix = IN_C64 && (p.first_frame_index + frame == 0);
          static_for<0, 32>([&](auto ic) {
            constexpr int kc = decltype(ic)::value;
            float dmx = db[kc], dmn = db[kc];
            if (nanfix && dmx != dmx) { dmx = -500.f; dmn = 500.f; }            // _nan_safe, first frame
            if constexpr ((HOLD & 1) != 0) hmax[kc] = hw_max(hmax[kc], dmx);    // np.fmax: NaN ignored
            if constexpr ((HOLD & 2) != 0) hmin[kc] = hw_min(hmin[kc], dmn);
          });
        }
      }
    }
    TDSA_STAMP(11);
  }

#ifdef TDSA_TIMELINE
  __syncthreads();
  if (p.dbg != nullptr && blockIdx.x == 0)
    for (int i = tid; i < 8 * 8 * 16; i += C::WGT) p.dbg[i] = tl[i];
#endif
  if constexpr (HOLD != 0) {
    const long long prow = ((long long)blockIdx.x * FPW + slot) * N;
    static_for<0, 32>([&](auto ic) {
      constexpr int kc = decltype(ic)::value;
      if constexpr ((HOLD & 1) != 0) (p.part_max + prow + (kc ^ 16) * SG)[t] = hmax[kc];
      if constexpr ((HOLD & 2) != 0) (p.part_min + prow + (kc ^ 16) * SG)[t] = hmin[kc];
    });
  }
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
  const int by_waves = 2 * 4 * 64 / C::WGT;                   // kernel is built for 2 waves per SIMD
  const int wg_per_cu = per_cu < by_waves ? per_cu : by_waves;
  const int units = (n_frames + C::FPW - 1) / C::FPW;
  int grid = num_cu * wg_per_cu;
  if (grid > units) grid = units;
  if (grid < 1) grid = 1;
  g.grid = grid;
  return g;
}

template <int LOG2N, bool IN_C64, int HOLD>
inline hipError_t launch_one(const SpecParams& p, const LaunchGeom& g, hipStream_t s) {
  auto k = spectrum_kernel<LOG2N, IN_C64, HOLD>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(g.lds_bytes));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(g.grid), dim3(g.block), g.lds_bytes, s, p);
  return hipGetLastError();
}

template <int LOG2N>
hipError_t launch_n(int in_c64, const SpecParams& p, const LaunchGeom& g, hipStream_t s) {
  const int hold = p.out_lin == nullptr ? (p.hold_flags & 3) : 0;
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

}  // namespace tdsa
