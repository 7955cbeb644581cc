// tdsa_trace.hip - trace-domain kernels around the frame kernel:
//   avg_scan      : TraceAverager recurrence over the frames of a batch, one bin per thread, float64
//                   state (utils/signal_processing.py:35-61) + dB + cal offset + tare + hold
//   frame_sums / dc_track : the HackRF DC tracker for dc_alpha < 1 (hackrf_samples.py:360-365)
//   trace_update  : DataProcessor's per-frame cal offset / tare / hold on a dB row handed in from the
//                   host (display_data_processor.py:317-395)
#include "tdsa_kernels.hpp"

namespace tdsa {

constexpr float k10Log10_2f = 3.01029995663981195214f;

// float max/min through integer atomics (IEEE-754 order trick); the state is initialised to -inf / +inf
__device__ __forceinline__ void atomic_fmax(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_fmin(float* addr, float v) {
  if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

// One thread per bin walks the batch in frame order (the recurrence is order dependent: SURVEY.md 7).
__global__ void __launch_bounds__(64) avg_scan_kernel(const AvgParams p) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= p.n) return;
  double buf = p.count_in > 0 ? p.state[k] : 0.0;
  int count = p.count_in;
  const double one_minus_alpha = 1.0 - 1.0 / double(p.avg_n);
  const float alpha_f = float(1.0 / double(p.avg_n));   // numpy: python float * float32 array -> float32
  const float tare = p.tare != nullptr ? p.tare[k] : 0.f;
  float hmax = p.state_max != nullptr ? p.state_max[k] : 0.f;
  float hmin = p.state_min != nullptr ? p.state_min[k] : 0.f;
  constexpr int U = 8;
  for (int f0 = 0; f0 < p.n_frames; f0 += U) {
    float lin[U];
#pragma unroll
    for (int u = 0; u < U; ++u) lin[u] = (f0 + u < p.n_frames) ? p.lin[(size_t)(f0 + u) * p.n + k] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (f0 + u >= p.n_frames) break;
      if (count == 0) {            // buffer is None: adopt the frame (signal_processing.py:47-50)
        buf = double(lin[u]);
        count = 1;
      } else if (p.mode == 1) {    // exp (:52-55)
        buf = buf * one_minus_alpha;
        buf += double(alpha_f * lin[u]);
      } else {                     // lin (:56-59)
        if (count < p.avg_n) ++count;
        buf += (double(lin[u]) - buf) / double(count);
      }
      float db = fmaf(k10Log10_2f, __builtin_amdgcn_logf(float(buf + double(p.log_floor))), p.cal_db) - tare;
      if (p.out_db != nullptr) p.out_db[(size_t)(f0 + u) * p.n + k] = db;
      hmax = fmaxf(hmax, db);
      hmin = fminf(hmin, db);
    }
  }
  p.state[k] = buf;
  if (p.state_max != nullptr) p.state_max[k] = hmax;
  if (p.state_min != nullptr) p.state_min[k] = hmin;
}

// ---- chunked scan for long batches -------------------------------------------------------------------
// Both recurrences are linear in the state with data-independent coefficients:
//   s_f = a_f * s_(f-1) + b_f(P_f),   exp: a = 1 - 1/n, b = float(1/n) * P   (first ever frame: a = 0, b = P)
//                                      lin: c_f = min(count_in + f + 1, n), a = 1 - 1/c_f, b = P / c_f
// so a chunk can be scanned from a zero state (pass 1), the chunk carries chained per bin (pass 2) and the
// chunk re-scanned from its true carry-in while the dB rows are written (pass 3).  Frames x bins parallel.
__device__ __forceinline__ void avg_coeff(const AvgParams& p, int f, double& a, double& bscale, bool& b_in_float) {
  const int seen = p.count_in + f;                 // frames folded in before this one
  if (p.mode == 1) {                               // exp
    if (seen == 0 && p.count_in == 0) { a = 0.0; bscale = 1.0; b_in_float = false; }
    else { a = 1.0 - 1.0 / double(p.avg_n); bscale = 1.0 / double(p.avg_n); b_in_float = true; }
  } else {                                         // lin
    const int c = seen + 1 < p.avg_n ? seen + 1 : p.avg_n;
    a = 1.0 - 1.0 / double(c); bscale = 1.0 / double(c); b_in_float = false;
  }
}
__device__ __forceinline__ double avg_step(const AvgParams& p, int f, double s, float lin) {
  // written exactly as the reference evaluates it (utils/signal_processing.py:47-59)
  const int seen = p.count_in + f;
  if (seen == 0) return double(lin);
  if (p.mode == 1) {
    s = s * (1.0 - 1.0 / double(p.avg_n));
    return s + double(float(1.0 / double(p.avg_n)) * lin);
  }
  const int c = seen + 1 < p.avg_n ? seen + 1 : p.avg_n;
  return s + (double(lin) - s) / double(c);
}

constexpr int kAvgChunk = 64;
// chunk c of the scan: fixed runs of 64 frames, or (wg_chunks > 0) wg_fold consecutive workgroup ranges of the frame
// kernel's persistent grid (or equal ranges of the batch: agg_w_local) - possibly empty
__device__ __forceinline__ void avg_chunk_bounds(const AvgParams& p, int c, int& f0, int& f1) {
  if (p.wg_chunks > 0) {
    // wg_fold consecutive workgroup ranges make one chunk (sizes whose grid has more than 256 workgroups)
    const int r = p.wg_fold > 1 ? p.wg_fold : 1;
    const int first = c * r, last = min(first + r, p.wg_chunks) - 1;
    int t;
    avg_wg_range(p, first, f0, t);
    avg_wg_range(p, last, t, f1);
    if (last < first) f1 = f0;
  } else {
    f0 = c * kAvgChunk;
    f1 = f0 + kAvgChunk < p.n_frames ? f0 + kAvgChunk : p.n_frames;
  }
}
typedef float avg_f4 __attribute__((ext_vector_type(4)));

// per-frame coefficients of one chunk, computed once per workgroup (one frame per thread; they hold float64
// divisions) and read back as LDS broadcasts
struct AvgChunkCoeff { double a[kAvgChunk], b[kAvgChunk]; int b_in_float[kAvgChunk]; };
__device__ __forceinline__ void avg_chunk_coeff(const AvgParams& p, int f0, int f1, AvgChunkCoeff& cc) {
  const int tid = threadIdx.x;
  if (f0 + tid < f1) {
    double a, bs; bool bf;
    avg_coeff(p, f0 + tid, a, bs, bf);
    cc.a[tid] = a; cc.b[tid] = bs; cc.b_in_float[tid] = bf ? 1 : 0;
  }
  __syncthreads();
}
template <int V>
__device__ __forceinline__ void avg_load_row(const float* row, float (&x)[V]) {
  if constexpr (V == 4) {
    const avg_f4 q = *reinterpret_cast<const avg_f4*>(row);
    x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
  } else {
    x[0] = row[0];
  }
}

// V bins per thread (4 when the rows allow 16-byte accesses), eight frames fetched ahead of the dependent chain
template <int V>
__global__ void __launch_bounds__(64) avg_chunk_local_kernel(const AvgParams p, double* carry) {
  __shared__ AvgChunkCoeff cc;
  const int c = blockIdx.y;
  const int f0 = c * kAvgChunk, f1 = f0 + kAvgChunk < p.n_frames ? f0 + kAvgChunk : p.n_frames;
  avg_chunk_coeff(p, f0, f1, cc);
  const int k = (blockIdx.x * 64 + threadIdx.x) * V;
  if (k >= p.n) return;
  double s[V];
#pragma unroll
  for (int v = 0; v < V; ++v) s[v] = 0.0;
  constexpr int U = 8;
  for (int fb = f0; fb < f1; fb += U) {
    float x[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (fb + u < f1) avg_load_row<V>(p.lin + (size_t)(fb + u) * p.n + k, x[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (fb + u < f1) {
        const double a = cc.a[fb + u - f0], bs = cc.b[fb + u - f0];
        const bool bf = cc.b_in_float[fb + u - f0] != 0;
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] = a * s[v] + (bf ? double(float(bs) * x[u][v]) : bs * double(x[u][v]));
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) carry[(size_t)c * p.n + k + v] = s[v];   // chunk result from a zero carry-in
}
// One thread per bin chains the chunk results in order.  The chunk's state multiplier A_c = prod a_f is the same for
// every bin: the 64 threads of a workgroup compute 64 chunks' worth of it cooperatively (one chunk each) into LDS
// instead of every thread redoing all of them (which made this small pass the longest of the three: 141 us of a
// 327 us C3-sized step), and the carries of eight chunks are fetched ahead of the dependent chain.
__global__ void __launch_bounds__(64) avg_chunk_chain_kernel(const AvgParams p, double* carry, int n_chunks) {
  __shared__ double As[64];
  const int tid = threadIdx.x, k = blockIdx.x * 64 + tid;
  const bool live = k < p.n;
  double s = (live && p.count_in > 0) ? p.state[k] : 0.0;
  for (int cb = 0; cb < n_chunks; cb += 64) {
    {
      const int c = cb + tid;
      double A = 1.0;
      if (c < n_chunks) {
        int f0, f1;
        avg_chunk_bounds(p, c, f0, f1);
        for (int f = f0; f < f1; ++f) { double a, bs; bool bf; avg_coeff(p, f, a, bs, bf); A *= a; }
      }
      As[tid] = A;
    }
    __syncthreads();
    const int nc = n_chunks - cb < 64 ? n_chunks - cb : 64;
    if (live) {
      for (int j0 = 0; j0 < nc; j0 += 8) {
        double loc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) loc[u] = (j0 + u < nc) ? carry[(size_t)(cb + j0 + u) * p.n + k] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (j0 + u < nc) {
            carry[(size_t)(cb + j0 + u) * p.n + k] = s;          // carry-in of chunk cb + j0 + u
            s = loc[u] + As[j0 + u] * s;
          }
        }
      }
    }
    __syncthreads();
  }
  if (live) p.state[k] = s;
}
// Chain for workgroup chunks (aggregates from the frame kernel, float32): hundreds of chunks per bin.  One thread per bin
// walking them in order is bound by latency twice over - few waves, and gfx9 counts loads and stores in one in-order
// vmcnt, so every batch of aggregate loads waits for the carry stores issued before it (93 us for 256 chunks of a
// C3-sized step).  Here wave q of a workgroup takes the q-th run of 64 chunks of its 64 bins: all 64 aggregates of a
// thread are fetched at once (64 registers), scanned from a zero state, the runs' (state, multiplier) pairs are folded
// through LDS, and every thread walks its run again from its true carry-in, now only storing.
template <int MAXQ>     // runs of 64 chunks per bin = waves per workgroup (4: up to 256 chunks)
__global__ void __launch_bounds__(64 * MAXQ) avg_wg_chain_kernel(const AvgParams p, double* carry) {
  __shared__ double Lq[16][64];
  __shared__ double Aq[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave index = run of 64 chunks: scalar
  const int r = p.wg_fold > 1 ? p.wg_fold : 1;                   // workgroup ranges folded into one chunk (<= 4)
  const int nsub = p.wg_chunks;
  const int nc = (nsub + r - 1) / r;
  const int k = min(int(blockIdx.x) * 64 + lane, p.n - 1);      // (a bin past the end redoes the last one)
  const double s_in = p.count_in > 0 ? p.state[k] : 0.0;
  // multipliers / validity flags of the workgroup ranges: the same for every bin, made once per call by
  // avg_weights_kernel; this wave's share is parked in LDS and read back eight at a time (fenced: left to itself the
  // compiler hoists all the broadcasts of a run into VGPRs)
  __shared__ double As[4 * 64];          // per chunk: product over its ranges
  __shared__ float Asub[4 * 64 * 4];     // per range (float: it only scales a float32 aggregate while ranges are folded)
  __shared__ float Vsub[4 * 64 * 4];
  const int c0 = q * 64;
  {
    double a = 1.0;
    for (int j = 0; j < r; ++j) {
      const int idx = (c0 + lane) * r + j;
      const bool in = idx < kAvgMaxWgChunks + 64;
      const double aj = in ? p.chunk_a[idx] : 1.0;
      Asub[(c0 + lane) * 4 + j] = float(aj);
      Vsub[(c0 + lane) * 4 + j] = in ? p.chunk_v[idx] : 0.f;
      a *= aj;
    }
    As[c0 + lane] = a;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // this run's entries are wave-private
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // aggregates: every load unconditional and issued before anything depends on one (a row that was never written is read
  // and dropped); SGPR buffer descriptors: the range offset is scalar, the bin offset the only per-lane address.  The r
  // ranges of a chunk are folded as they arrive: L <- a_j L + L_j.
  float loc[64];
#pragma unroll
  for (int u = 0; u < 64; ++u) loc[u] = 0.f;
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.agg), 0, int(unsigned(nsub) * unsigned(p.n) * 4u), 0x00020000);
  for (int j = 0; j < r; ++j) {
    float tmp[64];
#pragma unroll
    for (int u = 0; u < 64; ++u)
      tmp[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ar, unsigned(k) * 4u, unsigned(min((c0 + u) * r + j, nsub - 1)) * unsigned(p.n) * 4u, 0));
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float a8[8], v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a8[u] = Asub[(c0 + 8 * g + u) * 4 + j]; v8[u] = Vsub[(c0 + 8 * g + u) * 4 + j]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) loc[8 * g + u] = fmaf(a8[u], loc[8 * g + u], v8[u] != 0.f ? tmp[8 * g + u] : 0.f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = 0.0, ap = 1.0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    double a8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] = As[c0 + 8 * g + u];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = fma(a8[u], s, double(loc[8 * g + u]));
      ap *= a8[u];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  Lq[q][lane] = s;
  if (lane == 0) Aq[q] = ap;
  __syncthreads();
  s = s_in;
  for (int qq = 0; qq < q; ++qq) s = fma(Aq[qq], s, Lq[qq][lane]);
  const int nu = min(64, nc - c0);                               // scalar: chunks of this run
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(carry, 0, int(unsigned(nc) * unsigned(p.n) * 8u), 0x00020000);
  typedef unsigned cu32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    double a8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] = As[c0 + 8 * g + u];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (8 * g + u < nu) {
        const cu32x2 pk = {unsigned(__double_as_longlong(s)), unsigned(__double_as_longlong(s) >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(pk, cr, unsigned(k) * 8u, unsigned(c0 + 8 * g + u) * unsigned(p.n) * 8u, 0);   // carry-in of chunk c0 + 8 g + u
        s = fma(a8[u], s, double(loc[8 * g + u]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nu > 0 && c0 + 64 >= nc) p.state[k] = s;                   // the run that holds the last chunk leaves the new state
}

// Aggregates of equal ranges of the batch for the sizes whose frame kernel cannot form them (AvgParams::agg_w_local): the
// same float32 dot product, in frame order, as the frame kernel's (tdsa_spectrum_kernel.hpp, AGG) - V bins per thread,
// eight rows fetched ahead of the dependent chain.
template <int V>
__global__ void __launch_bounds__(64) avg_agg_local_kernel(const AvgParams p, float* agg) {
  const int c = blockIdx.y;
  int f0, f1;
  avg_wg_range(p, c, f0, f1);
  const int k = (blockIdx.x * 64 + threadIdx.x) * V;
  if (k >= p.n || f1 <= f0) return;
  float s[V];
#pragma unroll
  for (int v = 0; v < V; ++v) s[v] = 0.f;
  constexpr int U = 8;
  for (int fb = f0; fb < f1; fb += U) {
    float x[U][V], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = fb + u < f1 ? fb + u : f1 - 1;          // (past the end: the last row again, weight 0 - no branch)
      avg_load_row<V>(p.lin + (size_t)f * p.n + k, x[u]);
      w[u] = fb + u < f1 ? p.agg_w_local[f] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (fb + u < f1) {
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] = fmaf(w[u], x[u][v], s[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) agg[(size_t)c * p.n + k + v] = s[v];
}

template <int V>
__global__ void __launch_bounds__(64) avg_chunk_final_kernel(const AvgParams p, const double* carry) {
  const int k = (blockIdx.x * 64 + threadIdx.x) * V, c = blockIdx.y;
  if (k >= p.n) return;
  int f0, f1;
  avg_chunk_bounds(p, c, f0, f1);
  if (f1 <= f0) return;
  double s[V];
  float tare[V], hmax[V], hmin[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    s[v] = carry[(size_t)c * p.n + k + v];
    tare[v] = p.tare != nullptr ? p.tare[k + v] : 0.f;
    hmax[v] = -INFINITY; hmin[v] = INFINITY;
  }
  constexpr int U = V == 1 ? 8 : 4;      // rows fetched ahead of the dependent chain
  for (int fb = f0; fb < f1; fb += U) {
    float x[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (fb + u < f1) avg_load_row<V>(p.lin + (size_t)(fb + u) * p.n + k, x[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (fb + u < f1) {
        float db[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          s[v] = avg_step(p, fb + u, s[v], x[u][v]);
          db[v] = fmaf(k10Log10_2f, __builtin_amdgcn_logf(float(s[v] + double(p.log_floor))), p.cal_db) - tare[v];
          hmax[v] = fmaxf(hmax[v], db[v]);
          hmin[v] = fminf(hmin[v], db[v]);
        }
        if (p.out_db != nullptr) {          // rows are written once and not read again here: non-temporal
          float* o = p.out_db + (size_t)(fb + u) * p.n + k;
          if constexpr (V == 4) __builtin_nontemporal_store(avg_f4{db[0], db[1], db[2], db[3]}, reinterpret_cast<avg_f4*>(o));
          else __builtin_nontemporal_store(db[0], o);
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    if (p.state_max != nullptr) atomic_fmax(p.state_max + k + v, hmax[v]);
    if (p.state_min != nullptr) atomic_fmin(p.state_min + k + v, hmin[v]);
  }
}

int avg_scan_chunks(int n_frames) { return (n_frames + kAvgChunk - 1) / kAvgChunk; }

// one thread per chunk walks its frames backwards: w[f] = b_f * (product of a_g over the later frames of the chunk)
__global__ void __launch_bounds__(64) avg_weights_kernel(const AvgParams p, float* w, double* chunk_a, float* chunk_v) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= kAvgMaxWgChunks + 64) return;
  double tail = 1.0;
  float valid = 0.f;
  if (c < p.wg_chunks) {
    int f0, f1;
    avg_wg_range(p, c, f0, f1);     // ONE workgroup's range
    for (int f = f1 - 1; f >= f0; --f) {
      double a, bs; bool bf;
      avg_coeff(p, f, a, bs, bf);
      w[f] = float((bf ? double(float(bs)) : bs) * tail);
      tail *= a;
    }
    valid = f1 > f0 ? 1.f : 0.f;
  }
  chunk_a[c] = tail;        // product of the chunk's a_f; 1 for an empty chunk and past the end
  chunk_v[c] = valid;
}

hipError_t launch_avg_weights(const AvgParams& p, float* w, double* chunk_a, float* chunk_v, hipStream_t s) {
  hipLaunchKernelGGL(avg_weights_kernel, dim3((kAvgMaxWgChunks + 64 + 63) / 64), dim3(64), 0, s, p, w, chunk_a, chunk_v);
  return hipGetLastError();
}

hipError_t launch_avg_scan(const AvgParams& p, hipStream_t s, double* carry) {
  if (carry != nullptr && p.wg_chunks > 0) {
    // the frame kernel's workgroups have formed their chunks' aggregates (p.agg): chain them, re-scan the chunks
    const bool vec = (p.n % 4 == 0) && (reinterpret_cast<uintptr_t>(p.lin) % 16 == 0) &&
                     (p.out_db == nullptr || reinterpret_cast<uintptr_t>(p.out_db) % 16 == 0);
    if (p.wg_chunks > kAvgMaxWgChunks || size_t(p.wg_chunks) * p.n * 8 > 0xffffffffull) return hipErrorInvalidValue;
    const int fold = p.wg_fold > 1 ? p.wg_fold : 1;
    const int chunks = (p.wg_chunks + fold - 1) / fold;
    const int runs = (chunks + 63) / 64;
    if (runs > 4 || fold > 4) return hipErrorInvalidValue;        // (the host only takes this path for up to 256 chunks)
    if (p.agg_w_local != nullptr) {
      if (fold != 1) return hipErrorInvalidValue;
      float* agg = const_cast<float*>(p.agg);
      if (vec) hipLaunchKernelGGL(avg_agg_local_kernel<4>, dim3((p.n / 4 + 63) / 64, chunks), dim3(64), 0, s, p, agg);
      else hipLaunchKernelGGL(avg_agg_local_kernel<1>, dim3((p.n + 63) / 64, chunks), dim3(64), 0, s, p, agg);
    }
    hipLaunchKernelGGL(avg_wg_chain_kernel<4>, dim3((p.n + 63) / 64), dim3(64 * runs), 0, s, p, carry);
    if (p.state_only) return hipGetLastError();                   // the chain has left the new state: nothing else is wanted
    // short rows: four bins per thread leave the re-scan with too few waves (N = 512: 512 of them walking ~150 frames each,
    // 73 us for what N = 1024 does in 41) - one bin per thread there
    const bool wide = vec && size_t((p.n / 4 + 63) / 64) * size_t(chunks) >= 2048;     // (N = 2048 / 4096 either way: 78 / 81 against 80 / 81 us)
    if (wide) hipLaunchKernelGGL(avg_chunk_final_kernel<4>, dim3((p.n / 4 + 63) / 64, chunks), dim3(64), 0, s, p, carry);
    else hipLaunchKernelGGL(avg_chunk_final_kernel<1>, dim3((p.n + 63) / 64, chunks), dim3(64), 0, s, p, carry);
    return hipGetLastError();
  }
  if (carry != nullptr && p.n_frames > 2 * kAvgChunk) {
    const int n_chunks = (p.n_frames + kAvgChunk - 1) / kAvgChunk;
    // four bins per thread when every row starts on a 16-byte boundary (not the N/2+1-bin rows of the audio path)
    const bool vec = (p.n % 4 == 0) && (reinterpret_cast<uintptr_t>(p.lin) % 16 == 0) &&
                     (p.out_db == nullptr || reinterpret_cast<uintptr_t>(p.out_db) % 16 == 0);
    if (vec) {
      const dim3 grid((p.n / 4 + 63) / 64, n_chunks);
      hipLaunchKernelGGL(avg_chunk_local_kernel<4>, grid, dim3(64), 0, s, p, carry);
      hipLaunchKernelGGL(avg_chunk_chain_kernel, dim3((p.n + 63) / 64), dim3(64), 0, s, p, carry, n_chunks);
      hipLaunchKernelGGL(avg_chunk_final_kernel<4>, grid, dim3(64), 0, s, p, carry);
    } else {
      const dim3 grid((p.n + 63) / 64, n_chunks);
      hipLaunchKernelGGL(avg_chunk_local_kernel<1>, grid, dim3(64), 0, s, p, carry);
      hipLaunchKernelGGL(avg_chunk_chain_kernel, dim3((p.n + 63) / 64), dim3(64), 0, s, p, carry, n_chunks);
      hipLaunchKernelGGL(avg_chunk_final_kernel<1>, grid, dim3(64), 0, s, p, carry);
    }
    return hipGetLastError();
  }
  hipLaunchKernelGGL(avg_scan_kernel, dim3((p.n + 63) / 64), dim3(64), 0, s, p);
  return hipGetLastError();
}

// TraceAverager.process on one host-provided frame (sweep averager DataProcessor owns)
__global__ void __launch_bounds__(256) avg_frame_kernel(const double* lin, int n, double* state, int count_in,
                                                        int mode, int avg_n) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  double buf;
  if (count_in == 0) {
    buf = lin[k];
  } else if (mode == 1) {
    buf = state[k] * (1.0 - 1.0 / double(avg_n));
    buf += (1.0 / double(avg_n)) * lin[k];
  } else {
    int count = count_in < avg_n ? count_in + 1 : count_in;
    buf = state[k];
    buf += (lin[k] - buf) / double(count);
  }
  state[k] = buf;
}

hipError_t launch_avg_host_frame(const double* lin, int n, double* state, int count_in, int mode, int avg_n,
                                 hipStream_t s) {
  hipLaunchKernelGGL(avg_frame_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lin, n, state, count_in, mode,
                     avg_n);
  return hipGetLastError();
}

// ---- DC tracker (dc_alpha < 1) ----------------------------------------------------------------------
template <bool IN_C64>
__global__ void __launch_bounds__(256) frame_sums_kernel(const void* in, unsigned xor_mask,
                                                         long long frame_stride, int n, float2* sums) {
  __shared__ float red[8];
  const int f = blockIdx.x;
  const unsigned char* fb = static_cast<const unsigned char*>(in) + (long long)f * frame_stride;
  float sr = 0.f, si = 0.f;
  if constexpr (IN_C64) {
    const float2* x = reinterpret_cast<const float2*>(fb);
    for (int i = threadIdx.x; i < n; i += 256) { sr += x[i].x; si += x[i].y; }
  } else {
    // four samples (8 bytes) per lane and load; frame starts are only sample (2-byte) aligned
    struct __attribute__((packed, aligned(2))) U2 { unsigned x, y; };
    const U2* x = reinterpret_cast<const U2*>(fb);
    unsigned ui = 0, uq = 0;
#pragma unroll 8
    for (int i = threadIdx.x; i < n / 4; i += 256) {
      const U2 q = x[i];
      const unsigned a = q.x ^ xor_mask, b = q.y ^ xor_mask;
      ui = __builtin_amdgcn_udot4(a, 0x00010001u, ui, false);
      uq = __builtin_amdgcn_udot4(a, 0x01000100u, uq, false);
      ui = __builtin_amdgcn_udot4(b, 0x00010001u, ui, false);
      uq = __builtin_amdgcn_udot4(b, 0x01000100u, uq, false);
    }
    sr = float(ui); si = float(uq);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
  if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = sr; red[(threadIdx.x >> 6) * 2 + 1] = si; }
  __syncthreads();
  if (threadIdx.x == 0)
    sums[f] = float2{red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7]};
}

hipError_t launch_frame_sums(const void* in, int in_c64, unsigned xor_mask, long long frame_stride, int n,
                             int n_frames, float2* sums, hipStream_t s) {
  if (in_c64)
    hipLaunchKernelGGL(frame_sums_kernel<true>, dim3(n_frames), dim3(256), 0, s, in, xor_mask, frame_stride, n,
                       sums);
  else
    hipLaunchKernelGGL(frame_sums_kernel<false>, dim3(n_frames), dim3(256), 0, s, in, xor_mask, frame_stride, n,
                       sums);
  return hipGetLastError();
}

// dc_f = (1 - alpha) dc_(f-1) + alpha mean_f  (hackrf_samples.py:361-364) over the frames of a batch.  The recurrence is
// linear with a constant multiplier, so one workgroup scans it in three steps: every thread scans its own run of
// consecutive frames from a zero state, the 256 runs' (multiplier, result) pairs are scanned in LDS (eight doubling steps),
// every thread re-scans its run from its true carry-in and writes the per-frame subtract values.  (The first version
// walked all frames on one thread with a dependent global load per frame: 470 us for the 2440 frames of a C3 batch; the
// second chained the 256 runs on thread 0 and fetched one frame sum per step: 15 us.)  The scan runs in float64 on the
// float32 frame means; the state handed from call to call stays float32 like the reference's.
// parts > 1: sums[] holds the sums of hop-long blocks and frame f is blocks f .. f + parts - 1 (byte formats with a hop
// that divides the frame: integer-valued floats, their sum is the frame's exact sum whatever the order)
__global__ void __launch_bounds__(256) dc_track_kernel(const float2* sums, int parts, int n, int n_frames, float alpha, float in_off,
                                                        float in_scale, float2* dc_state, float2* dc_sub) {
  __shared__ double run_re[256], run_im[256], run_a[256];
  const int tid = threadIdx.x;
  const int per = (n_frames + 255) / 256;
  const int f0 = tid * per < n_frames ? tid * per : n_frames;
  const int f1 = f0 + per < n_frames ? f0 + per : n_frames;
  const double a = double(1.0f - alpha), b = double(alpha);
  const float inv_n = 1.0f / float(n);
  auto mean_of = [&](const float2 q, double& mr, double& mi) {
    mr = double((q.x * inv_n - in_off) * in_scale);
    mi = double((q.y * inv_n - in_off) * in_scale);
  };
  constexpr int U = 8;                       // frame sums fetched ahead of the dependent chain
  auto frame_sum = [&](int f) -> float2 {
    float2 q = sums[f];
    for (int j = 1; j < parts; ++j) { const float2 b2 = sums[f + j]; q.x += b2.x; q.y += b2.y; }
    return q;
  };
  double sr = 0.0, si = 0.0, A = 1.0;
  for (int fb = f0; fb < f1; fb += U) {
    float2 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = frame_sum(fb + u < f1 ? fb + u : f1 - 1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (fb + u < f1) {
        double mr, mi;
        mean_of(q[u], mr, mi);
        sr = a * sr + b * mr;
        si = a * si + b * mi;
        A *= a;
      }
    }
  }
  // inclusive scan of the runs: (A, L) o (A', L') = (A' A, L' + A' L), earlier run on the left
  run_re[tid] = sr; run_im[tid] = si; run_a[tid] = A;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    double pr = 0.0, pi = 0.0, pa = 1.0;
    if (tid >= d) { pr = run_re[tid - d]; pi = run_im[tid - d]; pa = run_a[tid - d]; }
    __syncthreads();
    if (tid >= d) {
      run_re[tid] = sr = sr + A * pr;
      run_im[tid] = si = si + A * pi;
      run_a[tid] = A = A * pa;
    }
    __syncthreads();
  }
  const double c0r = double(dc_state->x), c0i = double(dc_state->y);      // units of x
  // carry-in of run tid: runs 0 .. tid - 1 applied to the state the call started from
  double cr = c0r, ci = c0i;
  if (tid > 0) { cr = run_re[tid - 1] + run_a[tid - 1] * c0r; ci = run_im[tid - 1] + run_a[tid - 1] * c0i; }
  __syncthreads();                                                          // every thread has read the state
  if (tid == 255) *dc_state = float2{float(sr + A * c0r), float(si + A * c0i)};
  sr = cr; si = ci;
  for (int fb = f0; fb < f1; fb += U) {
    float2 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = frame_sum(fb + u < f1 ? fb + u : f1 - 1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (fb + u < f1) {
        double mr, mi;
        mean_of(q[u], mr, mi);
        sr = a * sr + b * mr;
        si = a * si + b * mi;
        dc_sub[fb + u] = float2{float(sr) / in_scale, float(si) / in_scale};      // residual on top of in_off, raw units
      }
    }
  }
}

hipError_t launch_dc_track(const float2* sums, int n, int n_frames, float alpha, float in_off, float in_scale,
                           float2* dc_state, float2* dc_sub, hipStream_t s, int parts) {
  hipLaunchKernelGGL(dc_track_kernel, dim3(1), dim3(256), 0, s, sums, parts < 1 ? 1 : parts, n, n_frames, alpha, in_off, in_scale,
                     dc_state, dc_sub);
  return hipGetLastError();
}

// ---- DataProcessor per-frame trace update -----------------------------------------------------------
__global__ void __launch_bounds__(256) trace_update_kernel(const TraceParams p) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= p.n) return;
  float db = p.db_in[k] + p.cal_db;                         // _apply_cal_offset (:317-327)
  if (p.tare_collect) {                                     // _apply_tare collecting (:335-343)
    const float lin = exp2f(db * (3.32192809488736234787f / 10.0f));   // 10^(dB/10)
    const float acc = p.tare_first ? lin : p.tare_acc[k] + lin;
    p.tare_acc[k] = acc;
    if (p.tare_finish)                                      // (:351-356)
    {
      // np.maximum(avg, 1e-30) PROPAGATES a NaN (fmaxf would drop it): a bin that saw a NaN during the run keeps a
      // NaN baseline, as in the reference
      const float avg = acc / float(p.tare_count);
      p.tare_base[k] = k10Log10_2f * __builtin_amdgcn_logf(avg < 1e-30f ? 1e-30f : avg);
    }
  }
  if (p.tare_active) db -= p.tare_base[k];                  // (:361-367)
  if (p.live != nullptr) p.live[k] = db;
  if (p.state_max != nullptr) {                             // _update_max_hold (:371-382)
    float m = p.max_first ? ((db != db) ? -500.f : db) : fmaxf(p.state_max[k], db);
    p.state_max[k] = m;
    if (p.max_copy != nullptr) p.max_copy[k] = m;
  }
  if (p.state_min != nullptr) {                             // _update_min_hold (:384-395)
    float m = p.min_first ? ((db != db) ? 500.f : db) : fminf(p.state_min[k], db);
    p.state_min[k] = m;
    if (p.min_copy != nullptr) p.min_copy[k] = m;
  }
}

hipError_t launch_trace_update(const TraceParams& p, hipStream_t s) {
  hipLaunchKernelGGL(trace_update_kernel, dim3((p.n + 255) / 256), dim3(256), 0, s, p);
  return hipGetLastError();
}

// Real-input (audio) path, datasources/audio_samples.py:121-132 of the reference.  Each real signal of a tick
// (the mono mix, left, right - both for stereo) goes through the frame kernel as z = signal + 0i, one transform per
// signal.  (Round 1 packed z = left + i*right into ONE transform and separated the channels afterwards: the float32
// rounding of the louder channel, ~1e-7 of ITS amplitude, then lands in the quieter one - 0.009 dB of error on a
// channel 60 dB down, 5.7 dB on one 80 dB down, garbage on a silent one; the reference transforms each channel on its
// own and has no such cross-talk.)
//   real_select : interleaved (L, R) float32 samples -> complex streams (mono: 0.5 (L + R) in float32 like the
//                 reference's `(left + right) * 0.5`)
//   real_fold   : one-sided power of a real signal from its full spectrum, X[k] = (Z[k] + conj Z[N-k]) / 2 (the
//                 imaginary part of z is zero, the mean of the two halves only averages rounding), non-DC /
//                 non-Nyquist bins doubled, PSD scale; row f of the output goes to row f * rows_per_frame + row
__global__ void __launch_bounds__(256) real_select_kernel(const float2* __restrict__ lr, size_t count, int channel,
                                                          float2* __restrict__ za, float2* __restrict__ zb) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < count; i += stride) {
    const float2 s = lr[i];
    if (channel == 3) {
      za[i] = float2{s.x, 0.f};
      zb[i] = float2{s.y, 0.f};
    } else {
      const float v = channel == 1 ? s.x : (channel == 2 ? s.y : (s.x + s.y) * 0.5f);
      za[i] = float2{v, 0.f};
    }
  }
}

hipError_t launch_real_select(const float2* lr, size_t count, int channel, float2* za, float2* zb, hipStream_t s) {
  size_t blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(real_select_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lr, count, channel, za, zb);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) real_fold_kernel(const float2* __restrict__ spec, int n, int n_frames,
                                                        int rows_per_frame, int row, float pscale,
                                                        float* __restrict__ lin) {
  const int nb = n / 2 + 1;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)n_frames * nb) return;
  const int f = int(idx / nb), k = int(idx - (long long)f * nb);
  const float2 zk = spec[(long long)f * n + k];
  const float2 zn = spec[(long long)f * n + ((n - k) & (n - 1))];
  const float2 X = float2{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
  const float dbl = (k == 0 || k == n / 2) ? 1.0f : 2.0f;   // power[1:-1] *= 2
  lin[((long long)f * rows_per_frame + row) * nb + k] = (X.x * X.x + X.y * X.y) * pscale * dbl;
}

hipError_t launch_real_fold(const float2* spec, int n, int n_frames, int rows_per_frame, int row, float pscale,
                            float* lin, hipStream_t s) {
  const long long total = (long long)n_frames * (n / 2 + 1);
  hipLaunchKernelGGL(real_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, spec, n, n_frames,
                     rows_per_frame, row, pscale, lin);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) lin_to_db_kernel(const float* __restrict__ lin, size_t count, float log_floor,
                                                        float cal_db, float* __restrict__ out_db) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < count; i += stride) out_db[i] = fmaf(k10Log10_2f, __builtin_amdgcn_logf(lin[i] + log_floor), cal_db);
}

hipError_t launch_lin_to_db(const float* lin, size_t count, float log_floor, float cal_db, float* out_db,
                            hipStream_t s) {
  size_t blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(lin_to_db_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lin, count, log_floor, cal_db, out_db);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) p[i] = v;
}

hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
  return hipGetLastError();
}

}  // namespace tdsa
