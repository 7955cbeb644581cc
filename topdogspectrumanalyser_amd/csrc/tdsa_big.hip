// tdsa_big.hip - frames that do not fit the LDS: N = 2^15 ... 2^20 points (BASELINE.json config C5 is the
// 2^20-point Welch average; HackrfSamplesDataSource.set_num_samples of the reference is unbounded,
// datasources/hackrf_samples.py:392-405).
//
// Four-step FFT over N = N1 * N2 with N2 = 16384 (the largest LDS-resident size) and N1 = N / 16384 = 2..64;
// n = n1*N2 + n2, k = k1 + N1*k2:
//   (1) cols_kernel<N1> : per n2: unpack + DC + window, N1-point DFT over n1 entirely in registers (the N1
//                         inputs of a column sit N2 samples apart: every load is a coalesced run along n2),
//                         times W_N^(n2*k1), written as complex64 rows Z[seg][k1][n2]  (2N B in, 8N B out)
//   (2) the frame kernel (tdsa_spectrum_kernel.hpp, ACC variant): every row Z[seg][k1][.] is one 16384-point
//                         frame; |X|^2 of the segments a workgroup takes of one k1 is summed in registers and
//                         leaves as that workgroup's own row P[k1][j][k2]               (8N B in, 4N/K B out)
//   (3) gather_kernel   : sum over j of P[k1][j][k2] -> natural bin order k = k1 + N1*k2, fftshift, float64
//   (4) finish_kernel   : mean / averager state -> 10*log10(. * scale + floor) + cal (- tare) -> dB row, hold
// Replaces np.fft.fft on a long frame + TraceAverager + _apply_cal_offset of the reference
// (hackrf_samples.py:370, utils/signal_processing.py:35-61, core/display_data_processor.py:317-327).
#include "tdsa_fft.hpp"
#include "tdsa_kernels.hpp"
#include "tdsa_spectrum_kernel.hpp"   // Cfg<14>, the LDS / half-wave exchange helpers and the radix networks of the frame kernel

namespace tdsa {

constexpr int kRowLog2 = 14, kRowN = 1 << kRowLog2;    // N2

struct BigColsParams {
  const unsigned char* in;   // interleaved int8/uint8 IQ, or complex64 samples (in_c64)
  int in_c64;
  long long seg_stride;      // bytes between segment starts
  const float* window;       // [N] window * input scale, natural order
  const float2* tw_seed;     // [NA - 1 + NB - 1][16384]: W_N^(n2 a), a = 1 .. NA-1, then W_N^(n2 8 b), b = 1 .. NB-1 (exact, rounded once)
  const float2* dc_sub;      // [K] per-segment DC estimate MINUS in_off, raw units (small: keeps float32 exact), or null
  float2* z;                 // [K][N1][N2]
  unsigned xor_mask;
  float in_off;
  unsigned in_valid;         // complex64 rows: samples from this index on are zeros and are not read (0: the whole row is there)
  BigWindow win;             // how the window reaches the column threads (WMODE of the kernel = win.mode)
  BigChirpPre pre;           // long chirp-z frames: raw samples in, unpacked and multiplied by window x chirp on load (pre.aw != null)
};


// Buffer-descriptor access for the column pass: every load / store of a thread is "lane offset + compile-time row
// offset" of a block-uniform base, so the row offsets travel as scalar offsets of one SGPR descriptor per array and no
// 64-bit per-lane address arithmetic is left (round 2's pointer version spent 626 of its 3293 VALU instructions per
// thread on v_add_co / v_addc pairs and 148 VGPRs: 3 waves per SIMD).
using brsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ brsrc_t big_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(bytes), 0x00020000);
}
typedef unsigned bu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned bu32x4 __attribute__((ext_vector_type(4)));

// A buffer store of more than 8 bytes whose data registers are overwritten by the next VALU instruction is a hazard on
// gfx950 that ROCm 7.2's hazard recognizer pads only when the store's soffset is not a register: in round 3, with
// `soffset = s4`, the compiler re-used v[78:81] for the next pair of rows right behind the store and rows came out
// corrupted run to run (1e-6 .. 6e-4 of the frame maximum, only with more than 256 workgroups in flight; two 8-byte
// stores were always right: tools/ubench/store_hazard.hip, profiles/r03_store_hazard.txt).  Rounds 3-4 kept the row
// offset in the VGPR offset so that the recognizer padded it; since round 5 the store and its wait states are ONE
// inline-asm block, whatever the compiler's recognizer thinks (tests/test_kernel_resources.py checks the ISA).
__device__ __forceinline__ void store16_pinned(bu32x4 data, brsrc_t r, unsigned voff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(data), "v"(voff), "s"(r) : "memory");
}

// One thread per column n2.  X[k1] of the column sits in v[bitrev(k1)] after the in-register DIF.
// WFLAT: one window value for every sample (BigWindow::flat: rectangular windows, the all-ones window of the chirp-z
// path) - no window loads; otherwise the window is a table [N], one 4-byte load per sample.  (Round 5 also evaluated
// cosine-sum windows - every window the reference builds is a0 - a1 cos(2 pi n / (N - 1)) - in the kernel, from three
// scalar row constants and two values per column, 2 FMAs per sample instead of the load: 249-252 us per 64-segment
// capture against 231-238 with the table, 50.6 against 47.4 at 8 segments, alternating in one process,
// profiles/r05_c5_experiments.txt - the pass has no VALU slots to spare; not kept.)
// DC: the per-segment DC estimate is subtracted (HackRF branch); the RTL branch / BASELINE config 5 has none and its
// instantiation skips the two subtractions per sample.
// PRE: long chirp-z frames - raw samples in, times window x chirp (BigChirpPre); its own instantiation, the Welch path's
// register count (3 waves per SIMD at 2^20 points) is not to move.
template <int LOG2N1, bool WFLAT, bool DC, bool PRE = false>
__global__ void __launch_bounds__(256) big_cols_kernel(const BigColsParams p) {
  constexpr int N1 = 1 << LOG2N1;
  const int n2 = blockIdx.x * 256 + threadIdx.x;
  const int seg = blockIdx.y;
  // (x - in_off) is exact in float32 (small integers / halves); the DC estimate is passed as its small
  // residual so that no 24-bit rounding of "128 + something" enters (at 2^20 points that rounding alone
  // left 3e-7 * A_max in the DC bin)
  const float off = p.in_off;
  float sub_re = 0.f, sub_im = 0.f;
  if constexpr (DC) { const c32 s = p.dc_sub[seg]; sub_re = s.x; sub_im = s.y; }
  const brsrc_t wr = big_rsrc(p.window, unsigned(N1) * kRowN * 4u);
  const unsigned wv = unsigned(n2) * 4u;
  auto sample = [&](float re_raw, float im_raw, float ww) -> c32 {
    if constexpr (DC) return c32{((re_raw - off) - sub_re) * ww, ((im_raw - off) - sub_im) * ww};
    else return c32{(re_raw - off) * ww, (im_raw - off) * ww};
  };
  // W_N^(n2*k1), k1 = a + 8b, is built from the seeds W^(n2*a) (a < 8) and W^(n2*8b) with at most one product.  The seeds
  // depend on the column only: they come from a per-plan table [seed][n2] (evaluated in double, rounded once) as
  // coalesced 8-byte loads.  Rounds 2-3 rebuilt each from a two-level table exp(-2 pi i m / N) = hi[m >> 10] lo[m & 1023]:
  // 28 gathers per thread whose 64 lanes hit up to 64 different lines each - in the texture unit they cost more cycles
  // than all of the column's sample, window and row accesses together - plus 14 complex products and one more rounding.
  // They are fetched FIRST and parked in LDS: gfx9 counts loads and stores in one in-order vmcnt, so a table load issued
  // between the row stores could only be waited for together with every store before it.
  constexpr int NA = N1 < 8 ? N1 : 8, NB = N1 / NA;
  __shared__ c32 seeds[NA + NB][256];
  static_for<1, NA>([&](auto ac) { constexpr int a = decltype(ac)::value; seeds[a][threadIdx.x] = p.tw_seed[(a - 1) * kRowN + n2]; });
  static_for<1, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; seeds[NA + b][threadIdx.x] = p.tw_seed[(NA - 1 + b - 1) * kRowN + n2]; });
  c32 v[N1];
  if constexpr (PRE) {
    // long chirp-z frames (tdsa_chirp.hip): step 1 - unpack, DC, window x chirp a[n] - folded into this load: the raw frame
    // is read instead of rows U[f][M] (never written).  Split plans (frames above 2^19 points): segment 2f + s is the half
    // [s H, s H + valid) of frame f.  Samples past `valid` meet the table's descriptor end: zeros.
    const int fr = p.pre.split_h ? seg >> 1 : seg, sh = p.pre.split_h ? (seg & 1) * p.pre.split_h : 0;
    const unsigned valid = p.pre.split_h ? unsigned((seg & 1) ? p.pre.n - p.pre.split_h : p.pre.split_h) : unsigned(p.pre.n);
    const unsigned char* rb = p.in + (long long)fr * p.seg_stride + (long long)sh * (p.pre.c64 ? 8 : 2);
    const brsrc_t ar = big_rsrc(p.pre.aw + sh, valid * 8u);
    float dcx = 0.f, dcy = 0.f;
    if (p.dc_sub != nullptr) { const c32 d = p.dc_sub[fr]; dcx = d.x; dcy = d.y; }
    const unsigned xm = p.xor_mask & 0xffffu;
    if (p.pre.c64) {
      const brsrc_t ir = big_rsrc(rb, valid * 8u);
      static_for<0, N1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const bu32x2 q = __builtin_amdgcn_raw_buffer_load_b64(ir, unsigned(n2) * 8u, unsigned(i) * kRowN * 8u, 0);
        const bu32x2 w = __builtin_amdgcn_raw_buffer_load_b64(ar, unsigned(n2) * 8u, unsigned(i) * kRowN * 8u, 0);
        v[i] = cmul(c32{__uint_as_float(q.x) - dcx, __uint_as_float(q.y) - dcy}, c32{__uint_as_float(w.x), __uint_as_float(w.y)});
      });
    } else {
      const brsrc_t ir = big_rsrc(rb, valid * 2u);
      static_for<0, N1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const unsigned u = unsigned(__builtin_amdgcn_raw_buffer_load_b16(ir, unsigned(n2) * 2u, unsigned(i) * kRowN * 2u, 0)) ^ xm;
        const bu32x2 w = __builtin_amdgcn_raw_buffer_load_b64(ar, unsigned(n2) * 8u, unsigned(i) * kRowN * 8u, 0);
        v[i] = cmul(c32{(float(u & 0xffu) - off) - dcx, (float((u >> 8) & 0xffu) - off) - dcy},
                    c32{__uint_as_float(w.x), __uint_as_float(w.y)});
      });
    }
  } else
  if (p.in_c64) {
    // (chirp-z rows: the padding behind the first in_valid samples is implied - the descriptor ends there and a load
    //  past its end returns zeros)
    const brsrc_t ir = big_rsrc(p.in + (long long)seg * p.seg_stride, (p.in_valid ? p.in_valid : unsigned(N1) * kRowN) * 8u);
    static_for<0, N1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const bu32x2 q = __builtin_amdgcn_raw_buffer_load_b64(ir, unsigned(n2) * 8u, unsigned(i) * kRowN * 8u, 0);
      float ww = p.win.flat;
      if constexpr (!WFLAT) ww = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, wv, unsigned(i) * kRowN * 4u, 0));
      v[i] = sample(__uint_as_float(q.x), __uint_as_float(q.y), ww);
    });
  } else {
    const brsrc_t ir = big_rsrc(p.in + (long long)seg * p.seg_stride, unsigned(N1) * kRowN * 2u);
    const unsigned xm = p.xor_mask & 0xffffu;
    // every load of the thread issued before the first conversion: the 2 N1 results land in the registers v[] will
    // occupy anyway, and the memory latency is paid once instead of once per batch of ~20 the scheduler keeps in flight.
    // The samples are read once: non-temporal (aux bit 1), so that the raw bytes do not displace Z from the caches
    unsigned ru[N1]; float rw[WFLAT ? 1 : N1];
    static_for<0, N1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      ru[i] = unsigned(__builtin_amdgcn_raw_buffer_load_b16(ir, unsigned(n2) * 2u, unsigned(i) * kRowN * 2u, 2));
    });
    if constexpr (!WFLAT) {
      static_for<0, N1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        rw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, wv, unsigned(i) * kRowN * 4u, 0));
      });
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, N1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const unsigned u = ru[i] ^ xm;
      float ww = p.win.flat;
      if constexpr (!WFLAT) ww = rw[i];
      v[i] = sample(float(u & 0xffu), float((u >> 8) & 0xffu), ww);
    });
  }
  dif<N1, 0, N1>(v);
  c32 lo[NA];
  static_for<1, NA>([&](auto ac) { constexpr int a = decltype(ac)::value; lo[a] = seeds[a][threadIdx.x]; });
  const brsrc_t zr = big_rsrc(p.z + (long long)seg * N1 * kRowN, unsigned(N1) * kRowN * 8u);
  // 16-byte stores: the two lanes of a pair (columns n2, n2 + 1) exchange one value per pair of rows (k1, k1 + 1), the
  // even lane then stores both columns of row k1, the odd lane both columns of row k1 + 1
  const bool odd = (threadIdx.x & 1) != 0;
  const unsigned zv = (unsigned(n2) & ~1u) * 8u + (odd ? unsigned(kRowN) * 8u : 0u);
  c32 xprev = c32{0.f, 0.f};
  static_for<0, NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    c32 hb = c32{1.f, 0.f};
    if constexpr (b > 0) hb = seeds[NA + b][threadIdx.x];
    static_for<0, NA>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      constexpr int k1 = a + 8 * b;
      c32 x = v[bitrev(k1, LOG2N1)];
      if constexpr (b == 0 && a > 0) x = cmul(x, lo[a]);
      else if constexpr (b > 0 && a == 0) x = cmul(x, hb);
      else if constexpr (b > 0) x = cmul(x, cmul(hb, lo[a]));
      if constexpr (N1 >= 2 && (k1 & 1) == 0) {
        xprev = x;                                  // row k1 (even): wait for row k1 + 1
      } else if constexpr (N1 >= 2) {
        // rows (k1 - 1, k1): the even lane keeps its row k1 - 1 value and sends its row k1 value, the odd lane the
        // other way round
        const float sx = odd ? xprev.x : x.x, sy = odd ? xprev.y : x.y;          // what the partner needs
        const float rx = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(sx), 0xB1, 0xf, 0xf, true));
        const float ry = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(sy), 0xB1, 0xf, 0xf, true));
        const float ox = odd ? x.x : xprev.x, oy = odd ? x.y : xprev.y;          // own value of the row this lane stores
        const bu32x4 pk = {__float_as_uint(odd ? rx : ox), __float_as_uint(odd ? ry : oy),
                           __float_as_uint(odd ? ox : rx), __float_as_uint(odd ? oy : ry)};
        store16_pinned(pk, zr, zv + unsigned(k1 - 1) * kRowN * 8u);
      }
    });
  });
}

// ---- row pass: 16384-point transforms of the rows Z[seg][k1][.], |X|^2 summed over the segments a workgroup takes ----
// The frame kernel's three radix passes (A x 32 x 32 with A = 16: tdsa_spectrum_kernel.hpp) on complex64 rows, as a
// kernel of its own since round 4.  Until then it was an instantiation of the frame kernel (ACC) that fetched the next
// row - 128 KB per CU - in one burst behind pass 1 and waited for it at the top of the next row (7.7 us per row against
// 4.7 us of arithmetic).  What the row needs is not an EARLY fetch but a SPREAD one: the CU's share of the fabric
// (~26 GB/s) has to flow for ~90 % of the row's period, and a burst of loads from 1024 threads at once is served worse
// than the same bytes in a trickle.  Measured per 64 segments, same box (profiles/r04_c5_experiments.txt): all eight
// 16-byte loads of a thread at the row top 125 us | 4 at the top + 4 behind pass 1: 122 | 2 / 2 / 2 / 2 over pass 1 and
// the middle pass's gather: 117 | one load at each of eight points through all three passes: 105.5 (shipped; 14 points
// are wired, the knobs TDSA_ROWS_FA / _FB choose among them: six other spreads 105 - 109).
// The loop body is branch-free (the row after the last one is the last one again) and carries no other vector-memory
// operation, so the only vmcnt waits the compiler places are the ones for the landing row.  123 VGPRs, no scratch.
struct BigRowsParams {
  const float2* z;           // [group][N1][16384] rows from the column pass
  long long seg_stride;      // bytes between segments (N1 * 16384 * 8)
  int group;                 // segments in this round
  int act;                   // workgroups per k1 in this launch: grid = N1 * act, workgroup b = k1 * act + j
  float* acc;                // P[(k1 * acc_split + j)][16384] partial power sums, one row per workgroup
  int acc_split;             // rows of P per k1
  int acc_add;               // 0: the row is overwritten (first round of a call), 1: added to
  const float2* tw;          // exp(-2 pi i m / 16384)
};

// the row pass of ONE work item: row k1, the j-th share of the round's segments (the whole kernel when workgroup b = k1 act + j
// takes item b; an item of the ticket queue of big_rows_gather_kernel below)
// COHERENT: the item's row of P leaves with agent-scope stores (through to memory: readable from another XCD inside the same
// launch without a write-back of the whole L2 - the fused launch below)
template <bool COHERENT = false>
__device__ __forceinline__ void big_rows_item(const BigRowsParams& p, const int k1, const int j) {
  using C = Cfg<kRowLog2>;
  constexpr int N = C::N, SG = C::SG, A = C::A, H = C::H;
  static_assert(A == 16 && C::M == 2 && !C::INL && C::NPASS == 3 && C::FPW == 1, "row pass is written for N = 16 x 32 x 32");
  constexpr int LH = ilog2(H);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c32* buf = reinterpret_cast<c32*>(smem);
  c32* twm = reinterpret_cast<c32*>(smem + C::DATA_BYTES);

  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int h = (tid >> 5) & 1;
  const int t = wave * 32 + (tid & 31);
  const bool odd_half = h != 0;
  const int chunk = (p.group + p.act - 1) / p.act;
  const int s0 = min(p.group, j * chunk), s1 = min(p.group, s0 + chunk);

  // element i of a pass lives at i + (i >> 5) (rows of 32 complex + 1 pad element)
  const int wr1_base = 33 * t + 16 * h;
  constexpr int rd_stride = SG + SG / 32;
  const int rd_base = t + (t >> 5);
  const int rdA = rd_base + h * rd_stride;
  const int wrM = rd_base + 8 * h * rd_stride;
  const int ka_mid = t % A;
  const int rd3A = (t / A) * rd_stride + ka_mid + h * A;
  const unsigned lane_in_off = unsigned(t) * 16u + unsigned(h) * 8192u;

  c32 twf_lo[3], twf_hi[4];
  static_for<0, 3>([&](auto ic) { constexpr int q = decltype(ic)::value; twf_lo[q] = p.tw[t * 2 * (q + 1)]; });
  static_for<0, 4>([&](auto ic) { constexpr int a = decltype(ic)::value; twf_hi[a] = p.tw[t * (8 * a + h)]; });
  if (tid < C::TWM) {
    const int b = tid / A, ka = tid % A;
    twm[tid] = p.tw[ka * b * (N / (32 * A))];
  }
  float pacc[16];
  static_for<0, 16>([&](auto ic) { pacc[decltype(ic)::value] = 0.f; });

  // row of segment s: 16384 complex64 values at z + s * seg_stride + k1 * 128 KiB; this thread's 16 values are the eight
  // 16-byte pieces at lane_in_off + i * 16 KiB: (row 2i + h of the 16 x 1024 view, columns 2t, 2t + 1)
  const unsigned char* zrow = reinterpret_cast<const unsigned char*>(p.z) + (long long)k1 * (N * 8);
  auto row_rsrc = [&](int s) { return make_rsrc(zrow + (long long)s * p.seg_stride, N * 8u); };
  constexpr int NL = 8;     // 16-byte loads per thread and row (sixteen 8-byte ones: 126 us against 107)
  u32x4 ld[8];
  auto row_load = [&](const rsrc_t& r, auto ic) {
    constexpr int i = decltype(ic)::value;     // (issued in the order pass 1's first butterflies consume them: no change)
    ld[i] = __builtin_amdgcn_raw_buffer_load_b128(r, lane_in_off, i * 16384, 2);   // last use of Z: non-temporal (plain: 144 us)
  };
  if (s0 < s1) {
    const rsrc_t r = row_rsrc(s0);
    static_for<0, NL>([&](auto ic) { row_load(r, ic); });
  }
  __syncthreads();        // twm is in place
  for (int s = s0; s < s1; ++s) {
    c32 v[16];            // v[jj * 8 + i] = sample (row 2i + h, column 2t + jj)
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      v[i] = c32{__uint_as_float(ld[i].x), __uint_as_float(ld[i].y)};
      v[8 + i] = c32{__uint_as_float(ld[i].z), __uint_as_float(ld[i].w)};
    });
    const int sn = s + 1 < s1 ? s + 1 : s;          // (past the end: the same row again, so that the loop has no branch)
    const rsrc_t rn = row_rsrc(sn);
    // The next row's eight 16-byte loads are spread over the row's work, one here, one there (fetch_at<position>):
    // positions 0 row top | 1 between pass 1's two radix-8 networks | 2 behind them | 3 middle of pass 1's combine |
    // 4 behind pass 1's LDS writes | 5 behind the middle pass's gather | 6 between its two twiddle batches | 7 behind its
    // radix network | 8 middle of its combine | 9 behind its LDS writes | 10 behind the last pass's gather | 11 behind its
    // pre-twiddle | 12 behind its radix network | 13 middle of its combine.   TDSA_ROWS_FA / _FB: loads per position, a
    // leading 1 (keeps the literal decimal) and 7 decimal digits each (positions 0-6 / 7-13).
#if !(defined(TDSA_DEV) && defined(TDSA_ROWS_FA))
#undef TDSA_ROWS_FA
#undef TDSA_ROWS_FB
#define TDSA_ROWS_FA 11010110
#define TDSA_ROWS_FB 11011010
#endif
    auto fetch_at = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
      constexpr auto cnt = [](int q) constexpr { int d = q < 7 ? TDSA_ROWS_FA : TDSA_ROWS_FB, e = q < 7 ? 6 - q : 13 - q; while (e-- > 0) d /= 10; return d % 10; };
      constexpr int lo = [&] { int x = 0; for (int q = 0; q < P; ++q) x += cnt(q); return x; }();
      constexpr int n = cnt(P);
      static_assert(P < 13 || lo + n == NL, "all of the row's loads must be placed");
      static_for<lo, lo + n>([&](auto ic) { row_load(rn, ic); });
    };
#define TDSA_FETCH(P) fetch_at(std::integral_constant<int, P>{})
    TDSA_FETCH(0);
    // ---- pass 1: two radix-8 DFTs per half-thread, combine across the lane pair -> radix 16 ---------------
    radix<8, 0, 16>(v);
    TDSA_FETCH(1);
    radix<8, 8, 16>(v);
    TDSA_FETCH(2);
    static_for<0, 8>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u == 4) TDSA_FETCH(3);
      constexpr int jj0 = u / H, k0 = u % H, jj1 = (u + 8) / H, k1r = (u + 8) % H;
      constexpr int re = jj0 * H + bitrev(k0, LH), ro = jj1 * H + bitrev(k1r, LH);
      swap_halves(v[re], v[ro]);
      combine_const<k0, A>(v[re], v[ro]);
      constexpr int li = jj0 * A + k0;
      lds_st(&buf[wr1_base + li], v[re]);
      lds_st(&buf[wr1_base + li + H], v[ro]);
    });
    TDSA_FETCH(4);
    __syncthreads();
    // ---- middle radix-32 pass, in place --------------------------------------------------------------
    static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; v[i] = lds_ld(&buf[rdA + i * 2 * rd_stride]); });
    TDSA_FETCH(5);
    int tw_o = h * A + ka_mid;
    asm volatile("" : "+v"(tw_o));
    static_for<0, 2>([&](auto bc) {
      constexpr int b0 = decltype(bc)::value * 4;
      if constexpr (b0 == 4) TDSA_FETCH(6);
      c32 tw8[8];
      static_for<0, 8>([&](auto ic) {
        constexpr int q = decltype(ic)::value;
        constexpr int i = b0 + (q & 3) + 8 * (q >> 2);
        tw8[q] = twm[tw_o + i * 2 * A];
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 4>([&](auto ic) { constexpr int q = decltype(ic)::value; bf_tw(v[b0 + q], v[b0 + q + 8], tw8[q], tw8[q + 4]); });
      __builtin_amdgcn_sched_barrier(0);
    });
    dit_rest<16, 0, 16>(v);
    TDSA_FETCH(7);
    static_for<0, 8>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u == 4) TDSA_FETCH(8);
      constexpr int re = bitrev(u, 4), ro = bitrev(u + 8, 4);
      swap_halves(v[re], v[ro]);
      combine32<u>(v[re], v[ro], odd_half);
      lds_st(&buf[wrM + u * rd_stride], v[re]);
      lds_st(&buf[wrM + (u + 16) * rd_stride], v[ro]);
    });
    TDSA_FETCH(9);
    __syncthreads();
    // ---- last radix-32 pass ----------------------------------------------------------------------------
    static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; v[i] = lds_ld(&buf[rd3A + 2 * i * A + ((2 * i * A) >> 5)]); });
    TDSA_FETCH(10);
    static_for<0, 3>([&](auto ic) { opaque(twf_lo[decltype(ic)::value]); });
    static_for<0, 4>([&](auto ic) { opaque(twf_hi[decltype(ic)::value]); });
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int a = i >> 2, q = i & 3;
      c32 te = twf_hi[a], to = twf_hi[a + 2];
      if constexpr (q != 0) { te = cmul(te, twf_lo[q - 1]); to = cmul(to, twf_lo[q - 1]); }
      bf_tw(v[i], v[i + 8], te, to);
    });
    TDSA_FETCH(11);
    dit_rest<16, 0, 16>(v);
    TDSA_FETCH(12);
    static_for<0, 8>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u == 4) TDSA_FETCH(13);
      constexpr int re = bitrev(u, 4), ro = bitrev(u + 8, 4);
      swap_halves(v[re], v[ro]);
      combine32<u, true>(v[re], v[ro], odd_half);
    });
    static_for<0, 16>([&](auto ic) {
      constexpr int q = decltype(ic)::value;
      const c32 X = v[bitrev(q, 4)];
      pacc[q] = fmaf(X.x, X.x, fmaf(X.y, X.y, pacc[q]));
    });
    __syncthreads();      // the next row's pass-1 writes must not overtake this row's last gathers
  }
  // the workgroup's share of its row's power sum leaves as its own row of P (a workgroup without segments contributes
  // zeros in the first round of a call and nothing afterwards)
  if (s1 > s0 || p.acc_add == 0) {
    float* arow = p.acc + (long long)(k1 * p.acc_split + j) * N + t + 8 * h * SG;
    static_for<0, 16>([&](auto ic) {
      constexpr int q = decltype(ic)::value;
      constexpr int kc = (q < 8 ? q : q + 8);
      float x = pacc[q];
      if (p.acc_add != 0) x += arow[kc * SG];
      if constexpr (COHERENT) __hip_atomic_store(&arow[kc * SG], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else arow[kc * SG] = x;
    });
  }
}

__global__ void __launch_bounds__(1024, 4) big_rows_kernel(const BigRowsParams p) {
  const int k1 = int(blockIdx.x) / p.act;
  big_rows_item<false>(p, k1, int(blockIdx.x) - k1 * p.act);
}

hipError_t launch_big_rows(const float2* z, long long seg_stride, int group, int n1, int act, float* acc, int acc_split,
                           int acc_add, const float2* tw, hipStream_t s) {
  using C = Cfg<kRowLog2>;
  static std::atomic<unsigned long long> attr_done{0};
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(big_rows_kernel), int(C::LDS_BYTES), attr_done);
  if (e != hipSuccess) return e;
  const BigRowsParams p{z, seg_stride, group, act, acc, acc_split, acc_add, tw};
  hipLaunchKernelGGL(big_rows_kernel, dim3(n1 * act), dim3(C::WGT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

// ---- the transposed four-step transform (complex in, complex out in natural order): second transform of the chirp-z
//      convolution of long frames that are not a power of two ----------------------------------------------------------
// Its input arrives in the layout the first transform's row pass leaves: T[k1][k2] = V[k1 + N1 k2].  With the output index
// m = m2 + 16384 m1:   Y[m] = sum_k1 W_N1^(k1 m1) W_M^(k1 m2) ( sum_k2 T[k1][k2] W_16384^(k2 m2) )
// - the rows T[k1][.] go through the frame kernel first (complex bins out), and this kernel finishes per column m2: times
// W_M^(k1 m2) (the column pass's seed table), an N1-point DFT over k1 in registers, rows Y[m1][m2] = natural order.  Every
// access is a coalesced run along m2; only the bins below out_valid are stored.
struct BigColsOutParams {
  const float2* r;           // [seg][N1][16384] row transforms R[k1][m2]
  long long seg_stride;      // bytes between segments, input and output (N1 * 16384 * 8)
  const float2* tw_seed;     // the column pass's table for this M
  float2* y;                 // [seg][N1 * 16384] natural order
  unsigned out_valid;        // bins wanted (0: all)
  BigChirpPost post;         // long chirp-z frames: the wanted bins leave as power / dB rows instead (post.n != 0; y is not written)
};

template <int LOG2N1>
__global__ void __launch_bounds__(256) big_cols_out_kernel(const BigColsOutParams p) {
  constexpr int N1 = 1 << LOG2N1;
  constexpr int NA = N1 < 8 ? N1 : 8, NB = N1 / NA;
  const int m2 = blockIdx.x * 256 + threadIdx.x;
  const int seg = blockIdx.y;
  const brsrc_t rr = big_rsrc(reinterpret_cast<const unsigned char*>(p.r) + (long long)seg * p.seg_stride, unsigned(N1) * kRowN * 8u);
  c32 v[N1];
  static_for<0, N1>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const bu32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rr, unsigned(m2) * 8u, unsigned(i) * kRowN * 8u, 0);
    v[i] = c32{__uint_as_float(q.x), __uint_as_float(q.y)};
  });
  c32 lo[NA], hi[NB];
  static_for<1, NA>([&](auto ac) { constexpr int a = decltype(ac)::value; lo[a] = p.tw_seed[(a - 1) * kRowN + m2]; });
  static_for<1, NB>([&](auto bc) { constexpr int b = decltype(bc)::value; hi[b] = p.tw_seed[(NA - 1 + b - 1) * kRowN + m2]; });
  static_for<1, N1>([&](auto kc) {                      // input k1 = a + 8 b times W_M^(k1 m2)
    constexpr int k1 = decltype(kc)::value;
    constexpr int a = k1 % 8, b = k1 / 8;
    if constexpr (b == 0) v[k1] = cmul(v[k1], lo[a]);
    else if constexpr (a == 0) v[k1] = cmul(v[k1], hi[b]);
    else v[k1] = cmul(v[k1], cmul(hi[b], lo[a]));
  });
  dif<N1, 0, N1>(v);                                     // Y[m1] in v[bitrev(m1)]
  if (p.post.n != 0) {
    // long chirp-z frames (tdsa_chirp.hip): step 4 - |X / M|^2, fftshift by n / 2, dB + cal - tare (or linear power rows for
    // the averager) - folded into these stores: bin k of frame f sits in this segment at m = k - s H (split plans: segment
    // 2f + s holds the bins [s H, s H + valid)) and lands at (k + n / 2) mod n of row f.  Hold traces: chirp_hold_kernel.
    const int n = p.post.n, half = n / 2;
    const int fr = p.post.split_h ? seg >> 1 : seg, kb = p.post.split_h ? (seg & 1) * p.post.split_h : 0;
    const int valid = p.post.split_h ? ((seg & 1) ? n - p.post.split_h : p.post.split_h) : n;
    float* lrow = p.post.out_lin ? p.post.out_lin + (long long)fr * n : nullptr;
    float* drow = p.post.out_db ? p.post.out_db + (long long)fr * n : nullptr;
    static_for<0, N1>([&](auto mc) {
      constexpr int m1 = decltype(mc)::value;
      const int m = m1 * kRowN + m2;
      if (m < valid) {
        int j = m + kb + half;
        if (j >= n) j -= n;
        const c32 x = v[bitrev(m1, LOG2N1)];
        const float xr = x.x * p.post.inv_m, xi = x.y * p.post.inv_m;
        const float pw = xr * xr + xi * xi;
        if (lrow != nullptr) {
          lrow[j] = pw * p.post.pscale;
        } else {
          float db;
          constexpr float kTenLog10Of2 = 3.01029995663981195214f;   // as chirp_post_kernel, operation for operation
          if (p.post.db_mode == 0) db = fmaf(2.0f * kTenLog10Of2, __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(pw) + p.post.log_floor), p.post.cal_db);
          else db = fmaf(kTenLog10Of2, __builtin_amdgcn_logf(fmaf(pw, p.post.pscale, p.post.log_floor)), p.post.cal_db);
          if (p.post.tare != nullptr) db -= p.post.tare[j];
          drow[j] = db;
        }
      }
    });
    return;
  }
  const unsigned nvalid = p.out_valid ? p.out_valid : unsigned(N1) * kRowN;
  // the descriptor ends behind the last wanted bin: stores past it are dropped by the hardware
  const brsrc_t yr = big_rsrc(reinterpret_cast<unsigned char*>(p.y) + (long long)seg * p.seg_stride, nvalid * 8u);
  static_for<0, N1>([&](auto mc) {
    constexpr int m1 = decltype(mc)::value;
    const c32 x = v[bitrev(m1, LOG2N1)];
    const bu32x2 pk = {__float_as_uint(x.x), __float_as_uint(x.y)};
    __builtin_amdgcn_raw_buffer_store_b64(pk, yr, unsigned(m2) * 8u, unsigned(m1) * kRowN * 8u, 0);
  });
}

template <int L>
static hipError_t cols_out_launch(const BigColsOutParams& p, int n_seg, hipStream_t s) {
  hipLaunchKernelGGL(big_cols_out_kernel<L>, dim3(kRowN / 256, n_seg), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_big_cols_out(int log2m, const float2* r, long long seg_stride, int n_seg, const float2* tw_seed, float2* y,
                               unsigned out_valid, hipStream_t s, const BigChirpPost* post) {
  const BigColsOutParams p{r, seg_stride, tw_seed, y, out_valid, post ? *post : BigChirpPost{}};
  switch (log2m - kRowLog2) {
    case 1: return cols_out_launch<1>(p, n_seg, s);
    case 2: return cols_out_launch<2>(p, n_seg, s);
    case 3: return cols_out_launch<3>(p, n_seg, s);
    case 4: return cols_out_launch<4>(p, n_seg, s);
    case 5: return cols_out_launch<5>(p, n_seg, s);
    case 6: return cols_out_launch<6>(p, n_seg, s);
    default: return hipErrorInvalidValue;
  }
}

// P[k1 * split + j][k2] (float: the row pass's per-workgroup partial power sums of this call, summed over j in
// double, in a fixed order: the result is reproducible bit for bit) -> natural bin order k = k1 + N1*k2,
// fftshift-ed (ks = k ^ N/2), through an LDS tile so that reads and writes are coalesced.
//   fin == null : dst[ks] = (add ? dst[ks] : 0) + S   (float64; the averager step follows separately)
//   fin != null : the same, then the finish arithmetic of big_finish_kernel on dst[ks] in the same thread
struct BigFinishParams;
__device__ __forceinline__ void big_finish_bin(const BigFinishParams& p, long long ks, double sum);

struct BigFinishParams {
  const double* src;   // [N] fftshift-ed: sum over `count` segments, or the averager state (count = 1)
  double* mean_out;    // [N] running mean (TraceAverager._buffer) or null
  int count;
  int db_mode;         // 0: 20log10(sqrt(mean)+floor), 1: 10log10(mean*scale+floor)
  float pscale, log_floor, cal_db;
  const float* tare;
  float* out_db;       // [N] or null
  float* hold_max;     // or null
  float* hold_min;
  int max_first, min_first;
};
__device__ __forceinline__ void big_finish_bin(const BigFinishParams& p, long long ks, double sum) {
  const double mean = sum / double(p.count);
  if (p.mean_out != nullptr) p.mean_out[ks] = mean;
  float db;
  if (p.db_mode == 0) db = 20.0f * log10f(sqrtf(float(mean)) + p.log_floor);
  else db = 10.0f * log10f(float(mean * double(p.pscale) + double(p.log_floor)));
  db += p.cal_db;
  if (p.tare != nullptr) db -= p.tare[ks];
  if (p.out_db != nullptr) p.out_db[ks] = db;
  if (p.hold_max != nullptr) p.hold_max[ks] = p.max_first ? ((db != db) ? -500.f : db) : fmaxf(p.hold_max[ks], db);
  if (p.hold_min != nullptr) p.hold_min[ks] = p.min_first ? ((db != db) ? 500.f : db) : fminf(p.hold_min[ks], db);
}

__global__ void __launch_bounds__(256) big_finish_kernel(const BigFinishParams p) {
  const long long ks = (long long)blockIdx.x * 256 + threadIdx.x;
  big_finish_bin(p, ks, p.src[ks]);
}

template <int LOG2N1>
__global__ void __launch_bounds__(256) big_gather_kernel(const float* s, int split, double* dst, int add, BigFinishParams fin,
                                                         int fuse) {
  // tile: R rows k1 x 64 columns k2 per workgroup (1024 workgroups at 2^20 points: the partial rows are read with
  // `split` independent loads per element in flight - as 256 workgroups of all N1 rows the pass took 23 us)
  constexpr int N1 = 1 << LOG2N1, T = 64, R = N1 < 16 ? N1 : 16;
  __shared__ double tile[R][T + 1];
  const int k2b = blockIdx.x * T, k1b = blockIdx.y * R;
  for (int i = threadIdx.x; i < R * T; i += 256) {
    const int k1 = i / T, c = i % T;
    const float* q = s + (long long)(k1b + k1) * split * kRowN + k2b + c;
    double acc = 0.0;                                     // the row pass's workgroup partials of this k1, in a fixed order
    int j = 0;
    for (; j + 4 <= split; j += 4) {
      const float a0 = q[(long long)j * kRowN], a1 = q[(long long)(j + 1) * kRowN], a2 = q[(long long)(j + 2) * kRowN],
                  a3 = q[(long long)(j + 3) * kRowN];
      acc += double(a0); acc += double(a1); acc += double(a2); acc += double(a3);
    }
    for (; j < split; ++j) acc += double(q[(long long)j * kRowN]);
    tile[k1][c] = acc;
  }
  __syncthreads();
  constexpr long long half = (long long)N1 * kRowN / 2;
  for (int i = threadIdx.x; i < R * T; i += 256) {
    const int c = i / R, k1 = i % R;
    const long long k = (long long)(k2b + c) * N1 + k1b + k1;
    const long long ks = k ^ half;
    const double x = tile[k1][c];
    const double sum = add ? dst[ks] + x : x;
    dst[ks] = sum;
    if (fuse) big_finish_bin(fin, ks, sum);
  }
}

// ---- row pass + gather + finish as ONE launch with a dependency-counted work queue (round 6; the round-5 verdict's
//      "tile queue, no device-wide barrier", for the part of the chain whose two kernels share a launch shape) -------------
// Tickets are drawn from one counter: 0 .. R-1 are the row items (k1, j), R .. R+G-1 gather tiles (four 16 x 64 tiles of the
// gather kernel per ticket, one per 256 threads), anything above ends the workgroup.  A workgroup that draws a gather ticket
// waits until rows_done has reached this launch's target: every row ticket has by then been DRAWN by a workgroup that is
// running and waits for nobody, so the wait ends whatever else shares the GPU (two plans on two streams, two ranks on one
// GPU: a grid barrier - every workgroup waiting for every other, started or not - deadlocks there).  The counters only
// grow; the host passes each launch its base values, so nothing is reset between launches.  The spin is bounded: a
// workgroup that gives up sets q->gave_up and leaves (tests read it), it never hangs the device.
struct BigQueue {
  unsigned long long ticket, rows_done, gave_up, pad;
};
struct BigGatherParams {
  const float* s;          // P rows
  int split;
  double* dst;
  int add;
  BigFinishParams fin;
  int fuse;
};

template <int LOG2N1>
__device__ __forceinline__ void big_gather_tile256(const BigGatherParams& g, double (*tile)[65], int tile_id, int t256) {
  constexpr int N1 = 1 << LOG2N1, T = 64, R = N1 < 16 ? N1 : 16;
  const int k2b = (tile_id % (kRowN / T)) * T, k1b = (tile_id / (kRowN / T)) * R;
  for (int i = t256; i < R * T; i += 256) {
    const int k1 = i / T, c = i % T;
    const float* q = g.s + (long long)(k1b + k1) * g.split * kRowN + k2b + c;
    double acc = 0.0;                                     // the row items' partials of this k1, in a fixed order
    // (plain loads, four in flight: the workgroup invalidated its caches when the rows were complete; agent-scope atomic
    //  loads instead are issued one at a time - sixteen memory round trips per thread, +25 us per capture)
    int j = 0;
    for (; j + 4 <= g.split; j += 4) {
      const float a0 = q[(long long)j * kRowN], a1 = q[(long long)(j + 1) * kRowN], a2 = q[(long long)(j + 2) * kRowN],
                  a3 = q[(long long)(j + 3) * kRowN];
      acc += double(a0); acc += double(a1); acc += double(a2); acc += double(a3);
    }
    for (; j < g.split; ++j) acc += double(q[(long long)j * kRowN]);
    tile[k1][c] = acc;
  }
  __syncthreads();                                        // (all four tile groups of the workgroup pass here together)
  constexpr long long half = (long long)N1 * kRowN / 2;
  for (int i = t256; i < R * T; i += 256) {
    const int c = i / R, k1 = i % R;
    const long long k = (long long)(k2b + c) * N1 + k1b + k1;
    const long long ks = k ^ half;
    const double x = tile[k1][c];
    const double sum = g.add ? g.dst[ks] + x : x;
    g.dst[ks] = sum;
    if (g.fuse) big_finish_bin(g.fin, ks, sum);
  }
  __syncthreads();
}

template <int LOG2N1>
__global__ void __launch_bounds__(1024, 4) big_rows_gather_kernel(const BigRowsParams p, const BigGatherParams g, BigQueue* q,
                                                                  unsigned long long ticket_base, unsigned long long rows_target,
                                                                  int n_row_items, int n_gather_tickets, int variant) {
  constexpr int N1 = 1 << LOG2N1, R = N1 < 16 ? N1 : 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ long long s_ticket;
  // `variant`: timing-only experiments behind tdsa_debug_knob big_fuse_gather = 1 + 2 x variant (results undefined): bit 0 no
  // acquire fence
  auto draw = [&]() -> long long {
    __syncthreads();
    if (threadIdx.x == 0)
      s_ticket = (long long)(__hip_atomic_fetch_add(&q->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_base);
    __syncthreads();
    return s_ticket;
  };
  // phase 1: row items while there are any (two loops, not one with a branch: as one loop the gather's addresses were
  // hoisted above the row pass and spilled - 30 dwords reloaded behind vmcnt(0) in every row)
  long long t = draw();
  while (t < n_row_items) {
    const int k1 = int(t) / p.act;
    big_rows_item<true>(p, k1, int(t) - k1 * p.act);
    __syncthreads();                                    // every thread's agent-scope stores of the item's row of P have been acknowledged ...
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&q->rows_done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before it counts
    // (a release fence here - __threadfence - is a write-back of the XCD's whole L2 per wave: it made the launch 200 us longer)
    t = draw();
  }
  // phase 2: gather tickets; all row tickets have been drawn by running workgroups, so this wait ends
  if (t < n_row_items + n_gather_tickets) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(&q->rows_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rows_target) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) {                     // seconds: something is wrong, leave instead of hanging the device
          __hip_atomic_fetch_add(&q->gave_up, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      // acquire at agent scope = invalidate this CU's L1 and the XCD's L2 of what they hold of P (the last capture's rows):
      // no write-back, unlike the release side
      if (!(variant & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  while (t < n_row_items + n_gather_tickets) {
    double (*tile)[65] = reinterpret_cast<double (*)[65]>(smem) + (threadIdx.x >> 8) * R;
    big_gather_tile256<LOG2N1>(g, tile, int(t - n_row_items) * 4 + int(threadIdx.x >> 8), int(threadIdx.x & 255));
    t = draw();
  }
}

template <int L>
static hipError_t rows_gather_launch(const BigRowsParams& p, const BigGatherParams& g, void* q, unsigned long long ticket_base,
                                     unsigned long long rows_target, int grid, int variant, hipStream_t s) {
  using C = Cfg<kRowLog2>;
  constexpr int N1 = 1 << L, R = N1 < 16 ? N1 : 16;
  static std::atomic<unsigned long long> attr_done{0};
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(big_rows_gather_kernel<L>), int(C::LDS_BYTES), attr_done);
  if (e != hipSuccess) return e;
  const int tiles = (kRowN / 64) * (N1 / R);
  hipLaunchKernelGGL(big_rows_gather_kernel<L>, dim3(grid), dim3(C::WGT), C::LDS_BYTES, s, p, g, static_cast<BigQueue*>(q),
                     ticket_base, rows_target, grid, (tiles + 3) / 4, variant);
  return hipGetLastError();
}

// one launch: the row pass of a round of `group` segments and gather + finish behind it; consumes n1 * act + tiles / 4 + grid
// tickets (*tickets_used) and n1 * act row counts of the queue
hipError_t launch_big_rows_gather(int log2n, const float2* z, long long seg_stride, int group, int n1, int act, float* acc,
                                  const float2* tw, double* dst, int add, double* mean_out, int count, int db_mode, float pscale,
                                  float log_floor, float cal_db, const float* tare, float* out_db, float* hold_max,
                                  float* hold_min, int max_first, int min_first, void* queue, unsigned long long ticket_base,
                                  unsigned long long rows_target, unsigned long long* tickets_used, int variant, hipStream_t s) {
  const BigRowsParams p{z, seg_stride, group, act, acc, act, 0, tw};
  const BigFinishParams fin{dst, mean_out, count, db_mode, pscale, log_floor, cal_db, tare, out_db, hold_max, hold_min,
                            max_first, min_first};
  const BigGatherParams g{acc, act, dst, add, fin, 1};
  const int grid = n1 * act;
  const int R = n1 < 16 ? n1 : 16;
  *tickets_used = (unsigned long long)grid + (unsigned long long)(((kRowN / 64) * (n1 / R) + 3) / 4) + (unsigned long long)grid;
  switch (log2n - kRowLog2) {
    case 1: return rows_gather_launch<1>(p, g, queue, ticket_base, rows_target, grid, variant, s);
    case 2: return rows_gather_launch<2>(p, g, queue, ticket_base, rows_target, grid, variant, s);
    case 3: return rows_gather_launch<3>(p, g, queue, ticket_base, rows_target, grid, variant, s);
    case 4: return rows_gather_launch<4>(p, g, queue, ticket_base, rows_target, grid, variant, s);
    case 5: return rows_gather_launch<5>(p, g, queue, ticket_base, rows_target, grid, variant, s);
    case 6: return rows_gather_launch<6>(p, g, queue, ticket_base, rows_target, grid, variant, s);
    default: return hipErrorInvalidValue;
  }
}

// ---- Welch partials across GPUs: the running mean of linear power leaves a plan as float32 / float64,
//      the partials of W plans are combined on ONE device: mean = sum_r c_r m_r / sum_r c_r in double, in rank order
//      (utils/signal_processing.py:56-59 is the running mean being reassembled), then the finish arithmetic -----------
template <typename T>
__global__ void __launch_bounds__(256) welch_export_kernel(const double* __restrict__ src, double div, T* __restrict__ dst,
                                                           long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = T(src[i] / div);
}

hipError_t launch_welch_export(const double* src, double div, void* dst, int as_f32, long long n, hipStream_t s) {
  const dim3 grid(unsigned((n + 255) / 256));
  if (as_f32) hipLaunchKernelGGL(welch_export_kernel<float>, grid, dim3(256), 0, s, src, div, static_cast<float*>(dst), n);
  else hipLaunchKernelGGL(welch_export_kernel<double>, grid, dim3(256), 0, s, src, div, static_cast<double*>(dst), n);
  return hipGetLastError();
}

struct WelchParts {
  const void* part[kWelchMaxParts];   // the partial means: this device's staging copy of a host slab, or the ranks' own
                                      // buffers read in place over xGMI (tdsa_peer_open)
  int n_parts;
  int count[kWelchMaxParts];          // segments behind each partial mean (0: not read)
};

template <typename T>
__global__ void __launch_bounds__(256) welch_combine_kernel(const WelchParts w, long long n, double* sum_out, BigFinishParams fin,
                                                            int native_db) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int r = 0; r < w.n_parts; ++r) {
    if (w.count[r] == 0) continue;
    // system-scope loads: a part may live in another GPU's memory, rewritten by that GPU's process between two launches of
    // this kernel - nothing this device cached of it earlier may be served again
    T m;
    if constexpr (sizeof(T) == 4) {
      const unsigned u = __hip_atomic_load(static_cast<const unsigned*>(w.part[r]) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      m = __uint_as_float(u);
    } else {
      const unsigned long long u = __hip_atomic_load(static_cast<const unsigned long long*>(w.part[r]) + i, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_SYSTEM);
      m = __longlong_as_double((long long)u);
    }
    acc += double(m) * double(w.count[r]);
  }
  if (sum_out != nullptr) sum_out[i] = acc;
  if (!native_db) {
    big_finish_bin(fin, i, acc);
  } else {   // LDS-resident sizes: the averager's state is the mean itself, dB as the scan's own rows form it
    const double mean = acc / double(fin.count);
    if (fin.mean_out != nullptr) fin.mean_out[i] = mean;
    float db = fmaf(k10Log10_2, __builtin_amdgcn_logf(float(mean + double(fin.log_floor))), fin.cal_db);
    if (fin.tare != nullptr) db -= fin.tare[i];
    if (fin.out_db != nullptr) fin.out_db[i] = db;
    if (fin.hold_max != nullptr) fin.hold_max[i] = fin.max_first ? ((db != db) ? -500.f : db) : fmaxf(fin.hold_max[i], db);
    if (fin.hold_min != nullptr) fin.hold_min[i] = fin.min_first ? ((db != db) ? 500.f : db) : fminf(fin.hold_min[i], db);
  }
}

hipError_t launch_welch_combine(const void* const* parts, const int* counts, int n_parts, int as_f32, long long n,
                                double* sum_out, double* mean_out, int total, int native_db, int db_mode, float pscale,
                                float log_floor, float cal_db, const float* tare, float* out_db, float* hold_max,
                                float* hold_min, int max_first, int min_first, hipStream_t s) {
  if (n_parts < 1 || n_parts > kWelchMaxParts) return hipErrorInvalidValue;
  WelchParts w{};
  w.n_parts = n_parts;
  for (int r = 0; r < n_parts; ++r) {
    w.part[r] = parts[r];
    w.count[r] = counts[r];
  }
  const BigFinishParams fin{nullptr, mean_out, total, db_mode, pscale, log_floor, cal_db, tare, out_db, hold_max, hold_min,
                            max_first, min_first};
  const dim3 grid(unsigned((n + 255) / 256));
  if (as_f32) hipLaunchKernelGGL(welch_combine_kernel<float>, grid, dim3(256), 0, s, w, n, sum_out, fin, native_db);
  else hipLaunchKernelGGL(welch_combine_kernel<double>, grid, dim3(256), 0, s, w, n, sum_out, fin, native_db);
  return hipGetLastError();
}

#ifdef TDSA_DEV
// developer hook (tdsa_debug_knob "big_pre_wgs"): k empty workgroups ahead of the column pass - the dispatcher hands
// workgroups to the eight XCDs round robin, so this moves the XCD every workgroup of the following launches lands on
__global__ void __launch_bounds__(64) xcd_shift_kernel() {}
hipError_t launch_xcd_shift(int wgs, hipStream_t s) {
  hipLaunchKernelGGL(xcd_shift_kernel, dim3(unsigned(wgs)), dim3(64), 0, s);
  return hipGetLastError();
}
#endif

// ---- shader clock: independent v_add_f32 chains, four waves per SIMD on every CU - the SIMD's saturated VALU rate -----
__global__ void __launch_bounds__(256) valu_clock_kernel(float* out, int iters, float seed) {
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = seed + float(j);
  const float c = seed * 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 64; ++r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r % 8]) : "v"(c));
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) sum += a[j];
  if (sum == 123.456f) out[blockIdx.x] = sum;     // (keeps the chains alive; never true)
}

hipError_t launch_valu_clock(float* scratch, int n_cu, int iters, hipStream_t s) {
  hipLaunchKernelGGL(valu_clock_kernel, dim3(unsigned(n_cu) * 4u), dim3(256), 0, s, scratch, iters, 1.0f);
  return hipGetLastError();
}

// ---- DC of long frames: exact sums (integers for byte formats), tracker in double ---------------------------
template <bool IN_C64>
__global__ void __launch_bounds__(256) big_sums_kernel(const void* in, unsigned xor_mask, long long frame_stride,
                                                       int n, double* sums) {   // sums[2f], sums[2f+1] (zeroed)
  __shared__ double red[8];
  const int f = blockIdx.y;
  const int per = n / gridDim.x, i0 = blockIdx.x * per;
  const unsigned char* fb = static_cast<const unsigned char*>(in) + (long long)f * frame_stride;
  double sr = 0.0, si = 0.0;
  if constexpr (IN_C64) {
    const float2* x = reinterpret_cast<const float2*>(fb);
#pragma unroll 8
    for (int i = i0 + threadIdx.x; i < i0 + per; i += 256) { sr += double(x[i].x); si += double(x[i].y); }
  } else {
    // eight samples (16 bytes) per lane and load, eight loads in flight (segment starts are only sample aligned).  The
    // first version fetched one sample per lane and step: 74 of the 320 us of a 64-segment capture with DC removal on.
    struct __attribute__((packed, aligned(2))) U4 { unsigned x, y, z, w; };
    const U4* x = reinterpret_cast<const U4*>(fb + (long long)i0 * 2);
    unsigned ui = 0, uq = 0;
    const int n16 = per / 8;
#pragma unroll 8
    for (int i = threadIdx.x; i < n16; i += 256) {
      const U4 q = x[i];
      const unsigned w4[4] = {q.x ^ xor_mask, q.y ^ xor_mask, q.z ^ xor_mask, q.w ^ xor_mask};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ui = __builtin_amdgcn_udot4(w4[j], 0x00010001u, ui, false);
        uq = __builtin_amdgcn_udot4(w4[j], 0x01000100u, uq, false);
      }
    }
    const uint16_t* xt = reinterpret_cast<const uint16_t*>(fb);      // (per is a multiple of 8 for every long size; kept general)
    for (int i = i0 + n16 * 8 + threadIdx.x; i < i0 + per; i += 256) {
      const unsigned u = (unsigned(xt[i]) ^ xor_mask) & 0xffffu;
      ui += u & 0xffu; uq += u >> 8;
    }
    sr = double(ui); si = double(uq);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
  if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = sr; red[(threadIdx.x >> 6) * 2 + 1] = si; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[2 * f], red[0] + red[2] + red[4] + red[6]);       // integer-valued doubles: exact in any order
    atomicAdd(&sums[2 * f + 1], red[1] + red[3] + red[5] + red[7]);
  }
}

// dc <- (1 - alpha) dc + alpha mean  (hackrf_samples.py:361-364), in double; dc_res[f] = dc / in_scale (raw units).
// The means are formed by all threads (one segment each: a global load and two divisions) and parked in LDS; thread 0 then
// only walks the recurrence (one thread doing everything, a dependent load per segment: 18 us for 64 segments).
__global__ void __launch_bounds__(256) big_dc_kernel(const double* sums, int n, int n_frames, double alpha, double in_off,
                                                      double in_scale, float2* dc_state, float2* dc_res) {
  __shared__ double mre[256], mim[256];
  double dr = double(dc_state->x), di = double(dc_state->y);
  for (int f0 = 0; f0 < n_frames; f0 += 256) {
    const int f = f0 + int(threadIdx.x);
    if (f < n_frames) {
      mre[threadIdx.x] = (sums[2 * f] / double(n) - in_off) * in_scale;
      mim[threadIdx.x] = (sums[2 * f + 1] / double(n) - in_off) * in_scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nb = n_frames - f0 < 256 ? n_frames - f0 : 256;
      for (int u = 0; u < nb; ++u) {
        dr = (1.0 - alpha) * dr + alpha * mre[u];
        di = (1.0 - alpha) * di + alpha * mim[u];
        mre[u] = dr; mim[u] = di;
      }
    }
    __syncthreads();
    if (f < n_frames) dc_res[f] = float2{float(mre[threadIdx.x] / in_scale), float(mim[threadIdx.x] / in_scale)};
    __syncthreads();
  }
  if (threadIdx.x == 0) *dc_state = float2{float(dr), float(di)};
}

// ---- host launchers ------------------------------------------------------------------------------------
template <int L>
static hipError_t cols_launch(const BigColsParams& p, int n_seg, hipStream_t s) {
  const dim3 grid(kRowN / 256, n_seg);
  const bool flat = p.win.mode == 2, dc = p.dc_sub != nullptr;
  if (p.pre.aw != nullptr) hipLaunchKernelGGL((big_cols_kernel<L, true, false, true>), grid, dim3(256), 0, s, p);
  else if (flat && dc) hipLaunchKernelGGL((big_cols_kernel<L, true, true>), grid, dim3(256), 0, s, p);
  else if (flat) hipLaunchKernelGGL((big_cols_kernel<L, true, false>), grid, dim3(256), 0, s, p);
  else if (dc) hipLaunchKernelGGL((big_cols_kernel<L, false, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((big_cols_kernel<L, false, false>), grid, dim3(256), 0, s, p);
  return hipGetLastError();
}
template <int L>
static hipError_t gather_launch(const float* src, int split, double* dst, int add, const BigFinishParams& fin, int fuse,
                                hipStream_t s) {
  constexpr int N1 = 1 << L, R = N1 < 16 ? N1 : 16;
  hipLaunchKernelGGL(big_gather_kernel<L>, dim3(kRowN / 64, N1 / R), dim3(256), 0, s, src, split, dst, add, fin, fuse);
  return hipGetLastError();
}

hipError_t launch_big_cols(int log2n, const void* in, int in_c64, long long seg_stride, int n_seg, const BigWindow& win,
                           const float2* tw_seed, const float2* dc_sub, float2* z,
                           unsigned xor_mask, float in_off, hipStream_t s, unsigned in_valid, const BigChirpPre* pre) {
  const BigColsParams p{static_cast<const unsigned char*>(in), in_c64, seg_stride, win.table, tw_seed, dc_sub, z, xor_mask,
                        in_off, in_valid, win, pre ? *pre : BigChirpPre{}};
  switch (log2n - kRowLog2) {
    case 1: return cols_launch<1>(p, n_seg, s);
    case 2: return cols_launch<2>(p, n_seg, s);
    case 3: return cols_launch<3>(p, n_seg, s);
    case 4: return cols_launch<4>(p, n_seg, s);
    case 5: return cols_launch<5>(p, n_seg, s);
    case 6: return cols_launch<6>(p, n_seg, s);
    default: return hipErrorInvalidValue;
  }
}

static hipError_t gather_dispatch(int log2n, const float* s_rows, int split, double* dst, int add, const BigFinishParams& fin,
                                  int fuse, hipStream_t s) {
  switch (log2n - kRowLog2) {
    case 1: return gather_launch<1>(s_rows, split, dst, add, fin, fuse, s);
    case 2: return gather_launch<2>(s_rows, split, dst, add, fin, fuse, s);
    case 3: return gather_launch<3>(s_rows, split, dst, add, fin, fuse, s);
    case 4: return gather_launch<4>(s_rows, split, dst, add, fin, fuse, s);
    case 5: return gather_launch<5>(s_rows, split, dst, add, fin, fuse, s);
    case 6: return gather_launch<6>(s_rows, split, dst, add, fin, fuse, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_big_gather(int log2n, const float* s_rows, int split, double* dst, int add, hipStream_t s) {
  return gather_dispatch(log2n, s_rows, split, dst, add, BigFinishParams{}, 0, s);
}

// gather + finish in one launch: dst (+)= S, then mean = dst / count -> dB row, hold (Welch and plain frames)
hipError_t launch_big_gather_finish(int log2n, const float* s_rows, int split, double* dst, int add, double* mean_out, int count,
                                    int db_mode, float pscale, float log_floor, float cal_db, const float* tare,
                                    float* out_db, float* hold_max, float* hold_min, int max_first, int min_first,
                                    hipStream_t s) {
  const BigFinishParams fin{dst, mean_out, count, db_mode, pscale, log_floor, cal_db, tare, out_db, hold_max, hold_min,
                            max_first, min_first};
  return gather_dispatch(log2n, s_rows, split, dst, add, fin, 1, s);
}

hipError_t launch_big_finish(const double* src, long long n, double* mean_out, int count, int db_mode, float pscale,
                             float log_floor, float cal_db, const float* tare, float* out_db, float* hold_max,
                             float* hold_min, int max_first, int min_first, hipStream_t s) {
  BigFinishParams p{src, mean_out, count, db_mode, pscale, log_floor, cal_db, tare, out_db, hold_max, hold_min,
                    max_first, min_first};
  hipLaunchKernelGGL(big_finish_kernel, dim3(unsigned(n / 256)), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_big_dc(const void* in, int in_c64, unsigned xor_mask, long long frame_stride, int n, int n_frames,
                         double alpha, double in_off, double in_scale, double* sums, float2* dc_state, float2* dc_res,
                         hipStream_t s) {
  hipError_t e = hipMemsetAsync(sums, 0, size_t(n_frames) * 2 * sizeof(double), s);
  if (e != hipSuccess) return e;
  // a workgroup per 16384 samples; a call of few frames (one per GUI tick) spreads each over four times as many
  const dim3 grid(n / (n_frames >= 8 ? 16384 : 4096), n_frames);
  if (in_c64) hipLaunchKernelGGL(big_sums_kernel<true>, grid, dim3(256), 0, s, in, xor_mask, frame_stride, n, sums);
  else hipLaunchKernelGGL(big_sums_kernel<false>, grid, dim3(256), 0, s, in, xor_mask, frame_stride, n, sums);
  hipLaunchKernelGGL(big_dc_kernel, dim3(1), dim3(256), 0, s, sums, n, n_frames, alpha, in_off, in_scale, dc_state,
                     dc_res);
  return hipGetLastError();
}

}  // namespace tdsa
