// Device-side building blocks of libperiod_hip.so (gfx950 / CDNA4 only).
//
// Everything here works on ONE signal window that is resident in LDS.  A window-projection
// (Periods.project, reference Periods.py:142-219) is a strided fold -> mean -> tile; it has
// no dense contraction, so there is no MFMA anywhere in this library.  The mappings are
//   * thread-per-residue, rows accumulated in order r = 0..R-1: bit-identical to the
//     reference's np.sum(cp, 0) (Periods.py:194).  Used wherever a projection is
//     materialised (bases, residual updates).
//   * wave-per-period passes (seg_group / wave_pass_single / wave_pass_multi): one 64-lane wavefront owns a base
//     period p, lanes own residues, consecutive lanes read consecutive LDS words
//     (conflict-free ds_read_b64).  The base residues are split at N mod p, which makes the row
//     count and every count weight wave-uniform; one pass can also yield the folds of 2p and
//     4p (row classes).  Eight periods are reduced across lanes with permlane swaps and DPP.
//     Used by the sweeps, where the reference's own norm (BLAS ddot inside np.linalg.norm)
//     has no defined order; the max|S| sweep keeps row-order sums (bit-identical).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ph {

constexpr int kWave = 64;
constexpr int kBlock = 256;      // 4 wavefronts: thread-per-residue / sequential kernels
constexpr int kBlockWide = 512;  // 8 wavefronts: sweep kernels (4 workgroups x 32 KiB windows per CU -> 32 waves/CU)
constexpr int kMaxWaves = 16;

constexpr unsigned kTrunc = 1u;
constexpr unsigned kOrth = 2u;
constexpr unsigned kSingle = 4u;

struct Tables {
  const int* orth_off;  // dense CSR by period: sub-periods p/f to project out (Periods.py:209-214)
  const int* orth_q;
  const int* fac_off;   // dense CSR by period: ordered proper divisors (Periods.py:548-549)
  const int* fac_q;
};

// ---------------------------------------------------------------- LDS tickets
typedef volatile __attribute__((address_space(3))) int* lds_int_ptr;

// The next value of an LDS counter for the calling lane.  `atomicAdd` by one lane goes through the compiler's atomic
// optimiser, which wraps it into a wave-wide aggregation (two v_mbcnt, s_bcnt1, a second exec mask, readfirstlane,
// v_add); ds_inc_rtn_u32 with the wrap-around bound at 2^32 - 1 is the same increment, is not touched by that pass, and
// -- unlike inline assembly -- stays known to the compiler as an outstanding LDS operation (it places the wait in front
// of the first use of the result, wherever the register allocator moves it).  Claim a ticket before a pass and look at
// it after the pass: the wait is free by then.
__device__ __forceinline__ int lds_ticket(int* counter) {
  return (int)__builtin_amdgcn_atomic_inc32((unsigned*)counter, 0xffffffffu, __ATOMIC_RELAXED, "workgroup");
}
// the ticket of lane 0 as a wave-uniform value
__device__ __forceinline__ int lds_ticket_value(int t) { return __builtin_amdgcn_readfirstlane(t); }

// ---------------------------------------------------------------- wavefront / block reductions
// Cross-lane moves for doubles on gfx950 without ds_bpermute (which costs LDS cycles and a per-lane address
// register per step -- registers the 64-VGPR kernels spill): DPP moves data inside a row of 16 lanes,
// v_permlane16_swap / v_permlane32_swap exchange odd-even rows / half-waves of two registers, ds_swizzle's
// bit mode covers the one remaining distance (lane ^ 4).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7 - i inside each group of 8
constexpr int kDppRor8 = 0x128;        // row_ror 8: lane i <- lane i ^ 8 inside a row of 16

// v[l ^ 4] (ds_swizzle bit mode: and 0x1f, or 0, xor 4)
__device__ __forceinline__ double swizzle_xor4_f64(double v) {
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x101F);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x101F);
  return __hiloint2double(hi, lo);
}

// All-reduce over the wavefront: half-waves and odd / even rows with permlane swaps, then DPP inside a row of 16 lanes
// (ror 8, half mirror, xor 2, xor 1) -- nothing goes through the LDS pipe.  (Round 3 paired lanes as the classic xor
// butterfly does and took the xor-4 step through ds_swizzle: an LDS-crossbar operation that queues behind the folds of
// every other wavefront of the CU, on the critical path of every workgroup reduction.)
template <bool MAXOP>
__device__ __forceinline__ double wave_allreduce(double v) {
  auto op = [](double x, double y) { return MAXOP ? fmax(x, y) : x + y; };
  {  // l ^ 32: swapping the upper half of one copy with the lower half of the other leaves [lower, lower] / [upper, upper]
    unsigned a0 = (unsigned)__double2loint(v), a1 = (unsigned)__double2hiint(v), b0 = a0, b1 = a1;
    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    v = op(__hiloint2double((int)r1[0], (int)r0[0]), __hiloint2double((int)r1[1], (int)r0[1]));
  }
  {  // l ^ 16: odd rows of one copy against even rows of the other
    unsigned a0 = (unsigned)__double2loint(v), a1 = (unsigned)__double2hiint(v), b0 = a0, b1 = a1;
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    v = op(__hiloint2double((int)r1[0], (int)r0[0]), __hiloint2double((int)r1[1], (int)r0[1]));
  }
  v = op(v, dpp_f64<kDppRor8>(v));
  v = op(v, dpp_f64<kDppHalfMirror>(v));  // (lane i + lane 7 - i of its group of 8: DPP, where round 3 went through ds_swizzle)
  v = op(v, dpp_f64<kDppXor2>(v));
  v = op(v, dpp_f64<kDppXor1>(v));
  return v;
}

__device__ __forceinline__ double wave_sum(double v) { return wave_allreduce<false>(v); }

__device__ __forceinline__ double wave_max(double v) { return wave_allreduce<true>(v); }

// A value that is identical in every lane, moved into scalar registers: long-lived wave-uniform doubles
// (norms, thresholds) otherwise hold two VGPRs each in kernels that sit at the 64-VGPR cap.
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// red[0] + red[1] + ... + red[nw - 1] in that order, with all kMaxWaves reads in flight before the first addition
// (a loop over the run-time wave count was compiled into nw dependent LDS round trips: ~1 us at 16 wavefronts).
// Slots >= nw are read but not used; the result is the value of the plain loop, bit for bit.
template <bool MAXOP = false>
__device__ __forceinline__ double red_combine(const double* red, int nw) {
  double v[kMaxWaves];
#pragma unroll
  for (int i = 0; i < kMaxWaves; ++i) v[i] = red[i];
  double t = MAXOP ? v[0] : 0.0 + v[0];
#pragma unroll
  for (int i = 1; i < kMaxWaves; ++i)
    if (i < nw) t = MAXOP ? fmax(t, v[i]) : t + v[i];
  return t;
}

// Sum over the workgroup; every thread gets the result.  `red` holds >= 2*kMaxWaves doubles.
// Partials are combined in wave order, so the result is identical in all threads.
__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  v = wave_sum(v);
  if ((tid & (kWave - 1)) == 0) red[tid >> 6] = v;
  __syncthreads();
  const double t = red_combine(red, nw);
  __syncthreads();
  return uniform_f64(t);
}

// block_sum without the trailing barrier: `red` must not be written again before the workgroup's next barrier
__device__ __forceinline__ double block_sum_once(double v, double* red) {
  const int tid = threadIdx.x;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  v = wave_sum(v);
  if ((tid & (kWave - 1)) == 0) red[tid >> 6] = v;
  __syncthreads();
  return uniform_f64(red_combine(red, nw));
}

// Workgroup argmax from the wave winners (value, period): largest value, lowest period among equals, period 0 = none.
// All reads in flight before the comparisons, results in scalar registers (identical in every lane).
__device__ __forceinline__ void red_argmax(const double* wbest, const int* wbestp, int nw, double& best, int& bestp) {
  double v[kMaxWaves];
  int p[kMaxWaves];
#pragma unroll
  for (int i = 0; i < kMaxWaves; ++i) {
    v[i] = wbest[i];
    p[i] = wbestp[i];
  }
  double b = 0.0;
  int bp = 0;
#pragma unroll
  for (int i = 0; i < kMaxWaves; ++i)
    if (i < nw && p[i] != 0 && (v[i] > b || (v[i] == b && p[i] < bp))) {
      b = v[i];
      bp = p[i];
    }
  best = uniform_f64(b);
  bestp = __builtin_amdgcn_readfirstlane(bp);
}

// sums of two values over the workgroup with one pair of barriers (every thread gets both)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
  const int tid = threadIdx.x;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  a = wave_sum(a);
  b = wave_sum(b);
  if ((tid & (kWave - 1)) == 0) {
    red[tid >> 6] = a;
    red[kMaxWaves + (tid >> 6)] = b;
  }
  __syncthreads();
  const double ta = red_combine(red, nw), tb = red_combine(red + kMaxWaves, nw);
  __syncthreads();
  a = uniform_f64(ta);
  b = uniform_f64(tb);
}

// Maximum over the workgroup (every thread gets it; NaN inputs are ignored by fmax, -inf if there is none).
__device__ __forceinline__ double block_max(double v, double* red) {
  const int tid = threadIdx.x;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  v = wave_max(v);
  if ((tid & (kWave - 1)) == 0) red[tid >> 6] = v;
  __syncthreads();
  const double t = red_combine<true>(red, nw);
  __syncthreads();
  return uniform_f64(t);
}

// ---------------------------------------------------------------- geometry of one fold
struct Fold {
  int p;      // period
  int rows;   // R = ceil(N / p)
  int nfull;  // residues j < nfull have R samples, the others R-1  (Periods.py:188-193)
  int trows;  // rows averaged in trunc mode: R if p | N else R-1     (Periods.py:178-184)
  __device__ __forceinline__ Fold(int N, int p_) : p(p_) {
    rows = (N + p_ - 1) / p_;
    const int shortn = rows * p_ - N;
    nfull = p_ - shortn;
    trows = shortn == 0 ? rows : rows - 1;
  }
  __device__ __forceinline__ int count(int j) const { return j < nfull ? rows : rows - 1; }
};

// Row-order column sum of residue j over `n` rows: ((x[j] + x[p+j]) + x[2p+j]) + ...
// Eight independent loads are issued ahead of their eight dependent adds, so the chain costs
// an add latency per row instead of an LDS round trip; the order of the adds is unchanged.
template <typename T>
__device__ __forceinline__ T column_sum(const T* __restrict__ xs, int j, int p, int n) {
  if (n <= 0) return T(0);
  const T* ptr = xs + j;
  T s = ptr[0];
  ptr += p;
  int r = 1;
  for (; r + 8 <= n; r += 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ptr[u * p];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
    ptr += 8 * p;
  }
  for (; r < n; ++r) {
    s += ptr[0];
    ptr += p;
  }
  return s;
}

// mean of residue j exactly as the reference forms it: non-trunc S/cnt (Periods.py:194),
// trunc np.mean over the complete rows (Periods.py:180-184).
template <typename T>
__device__ __forceinline__ T residue_mean(const T* __restrict__ xs, const Fold& f, int j, bool trunc) {
  const int n = trunc ? (f.trows < f.count(j) ? f.trows : f.count(j)) : f.count(j);
  const int div = trunc ? f.trows : f.count(j);
  const T s = column_sum(xs, j, f.p, n);
  return s / T(div);
}

// Means of a SHORT period with its rows dealt to several threads.  One thread per residue walking its column (the
// order of np.sum(cp, 0), Periods.py:194) leaves all but p threads idle behind a chain of N / p dependent additions:
// 3 us for an accepted period of small_to_large, 12 us for the period-3 winner of an m_best_gamma sweep.  Here
// thread (g, j), g < G = 2^k <= min(kSplitMax, width / p), adds rows g, g + G, ... of residue j in order, the G
// partial sums are combined in order of g (16 LDS reads in flight at a time), and msm[j] = S_j / cnt_j.  The sum is
// associated differently from the reference's -- within a few ulp of it; results that carry the 1e-10 bar (bases,
// powers of the sweeps) may use it, Periods.project itself (ph_project_batch) keeps the row order.
// `part` holds >= G p <= width elements and may be `msm` itself (one more barrier then); ends with a barrier: on
// return msm[0 .. p) is complete for every thread.  Threads >= width take no part.
constexpr int kSplitMax = 64;
constexpr int kSplitMinRows = 32;  // rows a residue must have before they are dealt to several threads (>= 16 per thread)
template <typename T>
__device__ __forceinline__ void split_row_means(const T* __restrict__ work, T* msm, T* part, int N, int p, int tid, int width) {
  const int rows = (N + p - 1) / p, nfull = p - (rows * p - N);  // Fold(N, p)
  // Residues with fewer than kSplitMinRows rows keep the row order (a chain that short costs nothing, and windows of a
  // few dozen samples -- where a last pick of m_best can hang on whether a residual is EXACTLY zero -- then project
  // bit for bit like the reference).
  int G = 1;
  if (rows >= kSplitMinRows)
    while (2 * G * p <= width && G < kSplitMax && rows >= 2 * G * (kSplitMinRows / 2)) G <<= 1;
  if (G == 1) {
    if (tid < width)
      for (int j = tid; j < p; j += width) {
        const int cnt = j < nfull ? rows : rows - 1;
        msm[j] = column_sum(work, j, p, cnt) / T(cnt);
      }
  } else {
    const int g = tid / p, j = tid - g * p;  // g < G for the threads that take part
    if (tid < G * p) {
      const int cnt = j < nfull ? rows : rows - 1;
      const int n = cnt > g ? (cnt - g + G - 1) / G : 0;  // rows g, g + G, ... below cnt
      part[tid] = column_sum(work + (size_t)g * p, j, G * p, n);
    }
    __syncthreads();
    T s = T(0);
    if (tid < p) {
      for (int k0 = 0; k0 < G; k0 += 16) {  // G is 2, 4, 8 or a multiple of 16
        T v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = part[(k0 + k < G ? k0 + k : 0) * p + tid];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += k0 + k < G ? v[k] : T(0);
      }
    }
    if (part == msm) __syncthreads();  // every partial has been read: the first p slots now take the means
    if (tid < p) msm[tid] = s / T(tid < nfull ? rows : rows - 1);
  }
  __syncthreads();
}

// dst[n] = mean[n mod p] for all n < N (tile, Periods.py:196-198); src and dst are LDS.
// Each thread only touches its own residue columns, so src == dst would also be legal.
template <typename T>
__device__ __forceinline__ void fold_mean_tile(const T* __restrict__ src, T* __restrict__ dst, int N,
                                               int p, bool trunc) {
  const Fold f(N, p);
  for (int j = threadIdx.x; j < p; j += blockDim.x) {
    const T m = residue_mean(src, f, j, trunc);
    const int cnt = f.count(j);
    for (int r = 0; r < cnt; ++r) dst[r * p + j] = m;
  }
}

// buf <- buf - project(buf, q) in place (one term of Periods.py:212-214).
template <typename T>
__device__ __forceinline__ void subtract_projection_inplace(T* __restrict__ buf, int N, int q, bool trunc) {
  const Fold f(N, q);
  for (int j = threadIdx.x; j < q; j += blockDim.x) {
    const T m = residue_mean(buf, f, j, trunc);
    const int cnt = f.count(j);
    for (int r = 0; r < cnt; ++r) buf[r * q + j] -= m;
  }
}

// Full Periods.project of the LDS window `src` into the LDS buffer `dst` (all flag
// combinations).  Ends with a barrier: dst is complete for every thread on return.
template <typename T>
__device__ __forceinline__ void project_lds(const T* __restrict__ src, T* __restrict__ dst, int N, int p,
                                            unsigned flags, const Tables& tb) {
  const bool trunc = flags & kTrunc;
  fold_mean_tile(src, dst, N, p, trunc);
  __syncthreads();
  if (flags & kOrth) {
    const int a = tb.orth_off[p], b = tb.orth_off[p + 1];
    for (int k = a; k < b; ++k) {
      subtract_projection_inplace(dst, N, tb.orth_q[k], trunc);
      __syncthreads();
    }
  }
}

// sum_n v[n]^2 over the workgroup (every thread gets it).
template <typename T>
__device__ __forceinline__ double block_sumsq(const T* __restrict__ v, int N, double* red) {
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const double t = (double)v[n];
    acc += t * t;
  }
  return block_sum(acc, red);
}

// periodic_norm (Periods.py:221-241) from a sum of squares: (sqrt(ss)/sqrt(N)) [/sqrt(p)].
__device__ __forceinline__ double periodic_norm_from_sq(double ss, int N, int p_div) {
  double v = sqrt(ss) / sqrt((double)N);
  if (p_div > 0) v = v / sqrt((double)p_div);
  return v;
}
// the same with sqrt(N) supplied (a scalar-register value of the caller: the compiler otherwise hoists the square root
// into a vector register pair that lives -- and is spilled -- across the whole kernel)
__device__ __forceinline__ double periodic_norm_from_sq_n(double ss, double sqrtN, int p_div) {
  double v = sqrt(ss) / sqrtN;
  if (p_div > 0) v = v / sqrt((double)p_div);
  return v;
}

// ---------------------------------------------------------------- tuned wave-per-period fold
// Per-period geometry, built once on the host for (N, p) and read with scalar loads: the hot
// loop has no integer or floating-point division.
struct PGeom {
  int rows;        // R = ceil(N / p)
  int nfull;       // residues j < nfull own R samples, the others R-1
  double w_full;   // 1 / R
  double w_short;  // 1 / (R-1)   (0 when R == 1)
};

// Where a window lives: in LDS (the normal case; volatile LDS-address-space reads, see
// fold_rows) or -- for windows longer than the LDS -- in a per-workgroup HBM workspace that the
// same code reads through ordinary global loads (served by L2).
template <typename T, bool LDS>
struct Win;
template <typename T>
struct Win<T, true> {
  typedef const volatile __attribute__((address_space(3))) T* ptr;
  static __device__ __forceinline__ ptr cast(const T* p) { return (ptr)p; }
};
template <typename T>
struct Win<T, false> {
  typedef const T* ptr;
  static __device__ __forceinline__ ptr cast(const T* p) { return p; }
};

constexpr int kPad = 256;  // LDS windows are followed by kPad zeroed elements: a 4-chunk group reads up to
                           // 255 elements past the last row of the window (see seg_group)

// s[c] += sum over `nrows` rows of the C chunks starting at `ptr` (row stride p).  U rows x C
// chunks of independent ds_read_b64 are issued back to back, then ONE lgkmcnt(0) (the scalar
// issue slot is as busy as the vector one here; the other waves of the CU cover the latency),
// then the adds.  The reads go through a volatile LDS-address-space pointer: the chunks of a
// row share one address register (immediate offsets 512 c) and hipcc cannot fuse chunk pairs
// into ds_read2st64_b64, which runs at half the LDS rate of ds_read_b64.  Row order per
// residue is preserved (bit-identical column sums).
template <typename T, int C, int U, bool LDS = true>
__device__ __forceinline__ void fold_rows(typename Win<T, LDS>::ptr ptr, int p, int nrows, double (&s)[C]) {
  int r = 0;
  for (; r + U <= nrows; r += U) {
    T v[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) v[u][c] = ptr[u * p + 64 * c];
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) s[c] += (double)v[u][c];
    ptr += U * p;
  }
  if (U > 1) {
    for (; r < nrows; ++r) {
      T v[C];
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] = ptr[64 * c];
#pragma unroll
      for (int c = 0; c < C; ++c) s[c] += (double)v[c];
      ptr += p;
    }
  }
}

// p < 64, max|S| mode: one lane per residue, rows in order (bit-identical sums).
template <typename T>
__device__ __forceinline__ double wave_partial_small_maxabs(const T* __restrict__ xs, int p, const PGeom& g,
                                                            int lane) {
  if (lane >= p) return 0.0;
  return fabs((double)column_sum(xs, lane, p, lane < g.nfull ? g.rows : g.rows - 1));
}

// p < 64: G = 64/p row groups fill the wavefront (lane = g p + j reads x[lane + r G p],
// contiguous), the G partial sums of a residue are combined with a shuffle tree.  Lane j < p
// returns S_p[j] (the other lanes hold partial sums).  Summation order differs from the
// reference's; only used where the reference's own order is undefined.
template <typename T>
__device__ __forceinline__ double wave_fold_small(const T* __restrict__ xs, int N, int p, int lane) {
  const int G = 64 / p;
  const int L = G * p;
  const int full = N / L;  // rows of L elements that exist for every lane < L
  const bool on = lane < L;
  const T* ptr = xs + (on ? lane : 0);
  double s0 = 0.0, s1 = 0.0;
  int r = 0;
  for (; r + 4 <= full; r += 4) {
    const T a = ptr[0], b = ptr[L], c = ptr[2 * L], d = ptr[3 * L];
    s0 += (double)a;
    s1 += (double)b;
    s0 += (double)c;
    s1 += (double)d;
    ptr += 4 * L;
  }
  for (; r < full; ++r) {
    s0 += (double)ptr[0];
    ptr += L;
  }
  const bool tail = on && (full * L + lane < N);
  const T tv = xs[tail ? full * L + lane : 0];
  double part = s0 + s1 + (tail ? (double)tv : 0.0);
  double tot = on ? part : 0.0;
  // binary tree over the row groups: lane l adds lane l + s p for s = G'/2, ..., 1 (G' = G rounded up
  // to a power of two); after log2(G') steps the lanes below p hold the totals
  int s = 1;
  while (s < G) s <<= 1;
  for (s >>= 1; s >= 1; s >>= 1) {
    const int src = lane + s * p;
    const double o = __shfl(tot, src & (kWave - 1), kWave);
    tot += (src < L) ? o : 0.0;
  }
  return tot;
}

template <typename T>
__device__ __forceinline__ double wave_partial_small(const T* __restrict__ xs, int N, int p, const PGeom& g,
                                                     int lane) {
  const double tot = wave_fold_small(xs, N, p, lane);
  const double w = (lane < g.nfull) ? g.w_full : g.w_short;
  return (lane < p) ? tot * tot * w : 0.0;
}

template <typename T, bool MAXABS, bool LDS>
__device__ __forceinline__ double wave_pass_single(const T* __restrict__ xs, int p, const PGeom g, int lane);

template <typename T, bool MAXABS, bool LDS = true>
__device__ __forceinline__ double wave_partial(const T* __restrict__ xs, int N, int p, const PGeom& g, int lane) {
  if (p >= 64) return wave_pass_single<T, MAXABS, LDS>(xs, p, g, lane);
  return MAXABS ? wave_partial_small_maxabs(xs, p, g, lane) : wave_partial_small(xs, N, p, g, lane);
}

// Cross-lane reduction of 8 consecutive periods with 10 shuffles instead of 48, "online":
// period k's per-lane partial is merged as soon as it is complete (level 1 pairs periods over
// lane bit 5, level 2 pairs of pairs over bit 4, level 3 over bit 3), so the period loop stays
// rolled -- one copy of the fold code in the instruction cache -- and at most three partials
// are pending.  Afterwards lane L holds the full value of period slot
// bit5(L) + 2 bit4(L) + 4 bit3(L) of the block.
// Cross-lane moves for doubles on gfx950 without the LDS crossbar (ds_bpermute):
// v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows of two registers,
// DPP moves data inside a row of 16 lanes.

// One butterfly level: `lo` and `hi` are per-lane partials of two different periods (or period
// groups).  Afterwards the lanes whose bit `mask` is clear hold lo summed over the lane pair
// (l, l ^ mask) and the lanes whose bit is set hold hi summed over the pair.
template <bool MAXABS>
__device__ __forceinline__ double butterfly_merge(double lo, double hi, int mask, int lane) {
  auto op = [](double x, double y) { return MAXABS ? fmax(x, y) : x + y; };
  if (mask == 32 || mask == 16) {
    unsigned l0 = (unsigned)__double2loint(lo), l1 = (unsigned)__double2hiint(lo);
    unsigned h0 = (unsigned)__double2loint(hi), h1 = (unsigned)__double2hiint(hi);
    if (mask == 32) {
      const auto r0 = __builtin_amdgcn_permlane32_swap(l0, h0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(l1, h1, false, false);
      l0 = r0[0]; h0 = r0[1]; l1 = r1[0]; h1 = r1[1];
    } else {
      const auto r0 = __builtin_amdgcn_permlane16_swap(l0, h0, false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(l1, h1, false, false);
      l0 = r0[0]; h0 = r0[1]; l1 = r1[0]; h1 = r1[1];
    }
    return op(__hiloint2double((int)l1, (int)l0), __hiloint2double((int)h1, (int)h0));
  }
  const bool up = lane & mask;  // mask == 8
  const double keep = up ? hi : lo;
  const double send = up ? lo : hi;
  return op(keep, dpp_f64<kDppRor8>(send));
}

// All-reduce inside every group of 8 consecutive lanes (three DPP steps).
template <bool MAXABS>
__device__ __forceinline__ double reduce8(double v) {
  auto op = [](double x, double y) { return MAXABS ? fmax(x, y) : x + y; };
  v = op(v, dpp_f64<kDppHalfMirror>(v));
  v = op(v, dpp_f64<kDppXor1>(v));
  v = op(v, dpp_f64<kDppXor2>(v));
  return v;
}

__device__ __forceinline__ int butterfly8_slot(int lane) {
  return ((lane >> 5) & 1) + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1) * 4;
}

// Visit ||P_p x||^2 (or max_s |S_p[s]| when MAXABS) for p = p_first, p_first + stride, ...
// <= p_hi.  consume(value, p) runs in the 8 lanes that own period p.  p_first and stride
// must be wave-uniform.
// DIRECT: every period is reduced over the wavefront on its own instead of through the 8-period butterfly -- dearer
// per period (an fp64 all-reduce is 12 cross-lane moves), but nothing is pending across the folds; it pays where a
// wavefront visits only a few periods between two barriers (k_small_to_large: 8.70 -> 8.19 ms per config-4 shard) and
// costs 10-25 % in the long sweeps.
template <typename T, bool MAXABS, bool LDS, bool DIRECT = false, typename F>
__device__ __forceinline__ void wave_sweep(const T* __restrict__ xs, int N, const PGeom* __restrict__ geom,
                                           int p_first, int p_hi, int stride, int lane, F&& consume) {
  if (DIRECT) {  // no state across periods: a full wave reduction per period (consume runs in every lane)
    for (int p = p_first; p <= p_hi; p += stride) {
      const double a = wave_partial<T, MAXABS, LDS>(xs, N, p, geom[p], lane);
      consume(MAXABS ? wave_max(a) : wave_sum(a), p);
    }
    return;
  }
  for (int pb = p_first; pb <= p_hi; pb += 8 * stride) {
    double l1 = 0.0, l2 = 0.0, l3 = 0.0, tot = 0.0;
    for (int k = 0; k < 8; ++k) {
      const int p = pb + k * stride;
      double a = 0.0;
      if (p <= p_hi) a = wave_partial<T, MAXABS, LDS>(xs, N, p, geom[p], lane);
      if ((k & 1) == 0) {
        l1 = a;
        continue;
      }
      a = butterfly_merge<MAXABS>(l1, a, 32, lane);
      if ((k & 2) == 0) {
        l2 = a;
        continue;
      }
      a = butterfly_merge<MAXABS>(l2, a, 16, lane);
      if ((k & 4) == 0) {
        l3 = a;
        continue;
      }
      tot = butterfly_merge<MAXABS>(l3, a, 8, lane);
    }
    tot = reduce8<MAXABS>(tot);
    const int p = pb + butterfly8_slot(lane) * stride;
    if (p <= p_hi) consume(tot, p);
  }
}

// ---------------------------------------------------------------- wavefront priorities of co-resident workgroups
// Two 16-wave workgroups share a CU in the window-pair kernels, and the instruction arbiter serves the OLDER
// wavefronts first: at config 2 (512 workgroups = one per slot) the workgroup placed first on a CU ran its ten sweeps
// in 1.8 ms while its neighbour starved and then needed the CU for another 0.8 ms on its own, with half the
// wavefronts to hide latencies behind (tools/clocks.py, -DPH_CLOCKS).  Progress-dependent priorities keep the two in
// step: the priority of the long, issue-bound phase is (-step) mod 3, so a workgroup one step behind its neighbour
// wins the arbitration two times out of three, and the short barrier-bound phases of a step run at priority 3 (they
// need few issue slots and every barrier in them waits for the slowest wavefront).  k_mbest_step1_pair 2.65 -> 2.44
// ms, m_best_gamma 3.34 -> 3.07 ms.  -DPH_WAVE_PRIO=0 leaves the priorities alone.
#ifndef PH_WAVE_PRIO
#define PH_WAVE_PRIO 1
#endif
__device__ __forceinline__ void prio_by_progress(int step) {
#if PH_WAVE_PRIO
  switch (((3 << 20) - step) % 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(2); break;
  }
#endif
}
__device__ __forceinline__ void prio_long_phase() {
#if PH_WAVE_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ void prio_short_phase() {
#if PH_WAVE_PRIO
  __builtin_amdgcn_s_setprio(3);
#endif
}

// ---------------------------------------------------------------- multi-period passes
// One pass over the window at base period p also yields the folds of 2p and 4p: row r of the
// p-fold belongs to residue j + (r mod m) p of the (m p)-fold, so keeping one accumulator per
// row parity class (m = 2) or per r mod 4 (m = 4) gives S_2p / S_4p exactly, and S_p as their
// sums, for the LDS traffic and the adds of a single period.  The host plans the passes so that
// every candidate period is produced exactly once (plan_passes in period_hip.hip).
struct PassPlan {
  int p;  // base period
  int m;  // 1, 2 or 4: the pass yields p, 2p (m >= 2) and 4p (m == 4); 0: p < 64, row-split path;
          // 8 + n (window-pair kernels only): p <= 64, row-split path yielding p, p/2, ..., p / 2^(n-1)
};

// ---------------------------------------------------------------- segmented passes
// For a base period p every count boundary of every produced period (p, 2p, 4p) falls on the
// same base residue: nfull_q - u p = N - k p for some integer k, and the only such value strictly
// inside (0, p) is c = N mod p.  Splitting the base residues into segment A = [0, c) and
// segment B = [c, p) therefore makes everything wave-uniform inside a segment: the number of
// rows a residue owns (R in A, R-1 in B) and the count weight of every produced residue class.
// No per-lane classification is left in the fold; only the last chunk of a segment masks the
// lanes past its end.
//
// seg_group: C chunks of 64 consecutive base residues starting at `base`; rows are added in
// blocks of U (U = M for M >= 2; row class u = r mod M); a[u][c] keeps row-order sums.
// acc layout: [0] period p, [1..2] period 2p (u = 0, 1), [3..6] period 4p (u = 0..3);
// wgt holds the matching scalar weights of this segment.
template <typename T, int M, int U, int C, bool MAXABS, bool MASK, bool LDS>
__device__ __forceinline__ void seg_group(typename Win<T, LDS>::ptr ptr, int p, int nrows, int nvalid, int lane,
                                          const double (&wgt)[7], double (&part)[3]) {
  static_assert(U % M == 0, "a row block must cover whole class cycles");
  double a[M][C];
  int r = 0;
  if (nrows >= U) {  // the first row block initialises the class sums (no zeroing, no adds; same row order)
    T v[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) v[u][c] = ptr[u * p + 64 * c];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < M; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        a[u][c] = (double)v[u][c];
#pragma unroll
        for (int w = u + M; w < U; w += M) a[u][c] += (double)v[w][c];
      }
    ptr += U * p;
    r = U;
  } else {
#pragma unroll
    for (int u = 0; u < M; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) a[u][c] = 0.0;
  }
  for (; r + U <= nrows; r += U) {
    T v[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) v[u][c] = ptr[u * p + 64 * c];
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): one wait per block, see fold_rows
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) a[u % M][c] += (double)v[u][c];
    ptr += U * p;
  }
  if (U > 1) {  // fewer than U rows left; their classes continue the cycle (r is a multiple of M)
    const int rem = nrows - r;
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
      if (u < rem) {
        // keep this a real (wave-uniform) branch: hipcc otherwise if-converts it into two
        // v_cndmask per accumulator on every group
        asm volatile("" ::: "memory");
        T v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = ptr[u * p + 64 * c];
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int c = 0; c < C; ++c) a[u % M][c] += (double)v[c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    // last group of the segment, holding exactly the chunks that are left: lanes past the end of the LAST one hold
    // garbage -- one class: its sum is zeroed; several classes: their squares run under the lane mask
    if (MASK && c == C - 1 && M == 1) a[0][c] = (64 * c + lane < nvalid) ? a[0][c] : 0.0;
    if (MASK && c == C - 1 && M > 1 && !(64 * c + lane < nvalid)) continue;
    if (M == 1) {  // M <= 2: plain sums of squares per class, weighted once per segment
      const double t = a[0][c];
      part[0] = MAXABS ? fmax(part[0], fabs(t)) : fma(t, t, part[0]);
    } else if (M == 2) {
      const double e = a[0][c], o = a[1 % M][c], t = e + o;
      if (MAXABS) {  // max |S| of p and of 2p (screening values: S_p = e + o is not a row-order sum)
        part[0] = fmax(part[0], fabs(t));
        part[1] = fmax(part[1], fmax(fabs(e), fabs(o)));
      } else {
        part[0] = fma(t, t, part[0]);
        part[1] = fma(e, e, part[1]);
        part[2] = fma(o, o, part[2]);
      }
    } else if (MAXABS) {
      const double e = a[0][c] + a[2 % M][c], o = a[1 % M][c] + a[3 % M][c], t = e + o;
      part[0] = fmax(part[0], fabs(t));
      part[1] = fmax(part[1], fmax(fabs(e), fabs(o)));
#pragma unroll
      for (int u = 0; u < M; ++u) part[2] = fmax(part[2], fabs(a[u][c]));
    } else {
      const double e = a[0][c] + a[2 % M][c], o = a[1 % M][c] + a[3 % M][c], t = e + o;
      part[0] = fma(t, t * wgt[0], part[0]);
      part[1] = fma(e, e * wgt[1], part[1]);
      part[1] = fma(o, o * wgt[2], part[1]);
#pragma unroll
      for (int u = 0; u < M; ++u) part[2] = fma(a[u][c], a[u][c] * wgt[3 + u], part[2]);
    }
  }
}

// Single-period norm passes over few rows (large p: NR = rows of the segment <= 6, half of all
// passes of the m_best sweep).  The rows are unrolled at compile time: all NR x C loads are in
// flight before the one wait, the sums start from row 0 and there is no accumulator
// initialisation, loop control or remainder handling per group (-15 % on these passes).
template <typename T, int NR, int C, bool MASK, bool LDS>
__device__ __forceinline__ void rows_group(typename Win<T, LDS>::ptr ptr, int p, int nvalid, int lane, double& part) {
  T v[NR][C];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < C; ++c) v[r][c] = ptr[r * p + 64 * c];
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    double t = (double)v[0][c];
#pragma unroll
    for (int r = 1; r < NR; ++r) t += (double)v[r][c];
    if (MASK && c == C - 1) t = (64 * c + lane < nvalid) ? t : 0.0;  // a masked group is cut in its LAST chunk only
    part = fma(t, t, part);
  }
}

// ---------------------------------------------------------------- passes with one dispatch (round 4)
// Per-lane partials of ||P_q x||^2 for q = p, 2p, 4p -- or of max_s |S_q[s]| (MAXABS) -- for a base period p >= 64.
// Through round 3 a pass walked its two segments in a loop: per segment a dozen scalar selects, the compare chain of a
// row-count switch and the flags of its fall-through -- on a CU whose ONE scalar unit serves 32 wavefronts
// (tools/micro/valu_rate.hip: 1.05 cycles per s_add_u32 per CU; the sweeps kept it 70-90 % busy).  A pass now dispatches
// once, on the row count of its base period, and runs both segments as straight-line code with compile-time indexed
// weights (tools/micro/pair_pass_bench.hip measures the float-pair twins: -27 % scalar instructions per pass).  Sums and
// their order are unchanged.
__device__ __forceinline__ double scalar_select_lt(int x, int y, double a, double b) {  // (x < y) ? a : b, wave-uniform
  double r;
  asm("s_cmp_lt_i32 %1, %2\n\ts_cselect_b64 %0, %3, %4" : "=s"(r) : "s"(x), "s"(y), "s"(a), "s"(b) : "scc");
  return r;
}

template <typename T, int NR, bool LDS>
__device__ __forceinline__ double wave_single_rows(typename Win<T, LDS>::ptr base, int p, int len, int lane) {
  constexpr int CG = NR <= 4 ? 4 : 2;
  double part = 0.0;
  const int whole = len >> 6;
  int c = 0;
  for (; c + CG <= whole; c += CG) rows_group<T, NR, CG, false, LDS>(base + 64 * c, p, 0, lane, part);
  for (; c < whole; ++c) {
    asm volatile("" ::: "memory");
    rows_group<T, NR, 1, false, LDS>(base + 64 * c, p, 0, lane, part);
  }
  const int rem = len & 63;
  if (rem) {
    asm volatile("" ::: "memory");
    rows_group<T, NR, 1, true, LDS>(base + 64 * whole, p, rem, lane, part);
  }
  return part;
}

// one segment of an M-class pass: columns [0, len) from `base` (lane included), nrows samples each
template <typename T, int M, int U, bool MAXABS, bool LDS>
__device__ __forceinline__ void wave_multi_segment(typename Win<T, LDS>::ptr base, int p, int len, int nrows, int lane,
                                                   const double (&wgt)[7], double (&part)[3]) {
  constexpr int CM = (M == 4) ? 2 : 4;
  const int whole = len >> 6;
  int c = 0;
  for (; c + CM <= whole; c += CM) seg_group<T, M, U, CM, MAXABS, false, LDS>(base + 64 * c, p, nrows, 64 * CM, lane, wgt, part);
  const int left = len - 64 * c;  // < 64 CM columns
  const typename Win<T, LDS>::ptr at = base + 64 * c;
  if (CM == 4 && left > 128) {
    if (left > 192) seg_group<T, M, U, (CM == 4 ? 4 : 1), MAXABS, true, LDS>(at, p, nrows, left, lane, wgt, part);
    else seg_group<T, M, U, (CM == 4 ? 3 : 1), MAXABS, true, LDS>(at, p, nrows, left, lane, wgt, part);
  } else if (left > 64) {
    seg_group<T, M, U, 2, MAXABS, true, LDS>(at, p, nrows, left, lane, wgt, part);
  } else if (left > 0) {
    seg_group<T, M, U, 1, MAXABS, true, LDS>(at, p, nrows, left, lane, wgt, part);
  }
}

// Per-lane partial of ||P_p x||^2 (MAXABS: of max_s |S_p[s]|, row-order sums), p >= 64
template <typename T, bool MAXABS, bool LDS>
__device__ __forceinline__ double wave_pass_single(const T* __restrict__ xs, int p, const PGeom g, int lane) {
  typedef typename Win<T, LDS>::ptr lds_ptr;
  const lds_ptr a = Win<T, LDS>::cast(xs) + lane, b = a + g.nfull;
  const int cut = g.nfull, rest = p - cut;
  double sa = 0.0, sb = 0.0;
  bool done = false;
  if (!MAXABS) {  // few rows: compile-time row count (norm passes only: the max|S| kernels spill)
    done = true;
    switch (g.rows) {
      case 2: sa = wave_single_rows<T, 2, LDS>(a, p, cut, lane); sb = wave_single_rows<T, 1, LDS>(b, p, rest, lane); break;
      case 3: sa = wave_single_rows<T, 3, LDS>(a, p, cut, lane); sb = wave_single_rows<T, 2, LDS>(b, p, rest, lane); break;
      case 4: sa = wave_single_rows<T, 4, LDS>(a, p, cut, lane); sb = wave_single_rows<T, 3, LDS>(b, p, rest, lane); break;
      case 5: sa = wave_single_rows<T, 5, LDS>(a, p, cut, lane); sb = wave_single_rows<T, 4, LDS>(b, p, rest, lane); break;
      case 6: sa = wave_single_rows<T, 6, LDS>(a, p, cut, lane); sb = wave_single_rows<T, 5, LDS>(b, p, rest, lane); break;
      default: done = false; break;
    }
  }
  if (!done) {
#ifndef PH_U1
#define PH_U1 2  // rows per load block of single-period passes (tuning knob)
#endif
    const double wgt[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double pa[3] = {0.0, 0.0, 0.0}, pb[3] = {0.0, 0.0, 0.0};
    wave_multi_segment<T, 1, PH_U1, MAXABS, LDS>(a, p, cut, g.rows, lane, wgt, pa);
    if (rest > 0) wave_multi_segment<T, 1, PH_U1, MAXABS, LDS>(b, p, rest, g.rows - 1, lane, wgt, pb);
    sa = pa[0];
    sb = pb[0];
  }
  return MAXABS ? fmax(sa, sb) : fma(sb, g.w_short, sa * g.w_full);  // segment A's product is rounded first (rounds 1-3: same bits)
}

// Per-lane partials for q = p, 2p (and 4p, M == 4), base period p >= 64: total[0], total[1], total[2]
template <typename T, int M, bool MAXABS, bool LDS>
__device__ __forceinline__ void wave_pass_multi(const T* __restrict__ xs, int p, const PGeom* __restrict__ geom, int lane,
                                                double (&total)[3]) {
  static_assert(M == 2 || M == 4, "two or four classes");
  typedef typename Win<T, LDS>::ptr lds_ptr;
  const PGeom g1 = geom[p], g2 = geom[2 * p], g4 = geom[(M == 4 ? 4 : 2) * p];
  const int cut = g1.nfull, rest = p - cut;
  const lds_ptr a = Win<T, LDS>::cast(xs) + lane;
  double wa[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, wb[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (!MAXABS) {
    wa[0] = g1.w_full;
    wb[0] = g1.w_short;
    wa[1] = g2.w_full;  // residue 0 is never short
    wa[3] = g4.w_full;
    wa[2] = scalar_select_lt(p, g2.nfull, g2.w_full, g2.w_short);
#pragma unroll
    for (int u = 0; u < 2; ++u) wb[1 + u] = scalar_select_lt(cut + u * p, g2.nfull, g2.w_full, g2.w_short);
    if (M == 4) {
#pragma unroll
      for (int u = 1; u < 4; ++u) wa[3 + u] = scalar_select_lt(u * p, g4.nfull, g4.w_full, g4.w_short);
#pragma unroll
      for (int u = 0; u < 4; ++u) wb[3 + u] = scalar_select_lt(cut + u * p, g4.nfull, g4.w_full, g4.w_short);
    }
  }
  if (M == 4) {  // weights (or maxima) applied per chunk, straight into the totals
    total[0] = total[1] = total[2] = 0.0;
    wave_multi_segment<T, M, M, MAXABS, LDS>(a, p, cut, g1.rows, lane, wa, total);
    if (rest > 0) wave_multi_segment<T, M, M, MAXABS, LDS>(a + cut, p, rest, g1.rows - 1, lane, wb, total);
    return;
  }
  double sa[3] = {0.0, 0.0, 0.0}, sb[3] = {0.0, 0.0, 0.0};
  wave_multi_segment<T, M, M, MAXABS, LDS>(a, p, cut, g1.rows, lane, wa, sa);
  if (rest > 0) wave_multi_segment<T, M, M, MAXABS, LDS>(a + cut, p, rest, g1.rows - 1, lane, wb, sb);
  if (MAXABS) {
    total[0] = fmax(sa[0], sb[0]);
    total[1] = fmax(sa[1], sb[1]);
  } else {
    total[0] = fma(sb[0], wb[0], sa[0] * wa[0]);  // segment A first, as in rounds 1-3: same bits
    total[1] = fma(sb[1], wb[1], fma(sb[2], wb[2], fma(sa[1], wa[1], sa[2] * wa[2])));
  }
  total[2] = 0.0;
}

// The online 8-period butterfly of wave_sweep as a state machine, for producers that deliver
// one to three periods at a time.  `k` is wave-uniform.
template <bool MAXABS = false>
struct Butterfly8 {
  double l1, l2, l3;
  int k, myp;
  __device__ __forceinline__ void reset() {
    k = 0;
    myp = 0;
    l1 = l2 = l3 = 0.0;
  }
  template <typename F>
  __device__ __forceinline__ void push(double a, int p, int lane, F&& consume) {
    if (butterfly8_slot(lane) == k) myp = p;
    if ((k & 1) == 0) {
      l1 = a;
    } else {
      a = butterfly_merge<MAXABS>(l1, a, 32, lane);
      if ((k & 2) == 0) {
        l2 = a;
      } else {
        a = butterfly_merge<MAXABS>(l2, a, 16, lane);
        if ((k & 4) == 0) {
          l3 = a;
        } else {
          const double tot = reduce8<MAXABS>(butterfly_merge<MAXABS>(l3, a, 8, lane));
          if (myp != 0) consume(tot, myp);
          myp = 0;
        }
      }
    }
    k = (k + 1) & 7;
  }
  template <typename F>
  __device__ __forceinline__ void flush(int lane, F&& consume) {
    while (k != 0) push(0.0, 0, lane, consume);
  }
};

// Norm sweep driven by a pass plan: visits ||P_q x||^2 of every period the passes
// plan[i_first], plan[i_first + stride], ... (< i_end) produce.  Periods arrive out of order;
// consume(value, q) runs in the 8 lanes that own q.
// Chains of periods up to 64 (PassPlan.m = 8 + levels, round 4 for the fp64 sweeps; ph_pair.h has the float-pair
// form): the row-split fold at L leaves S_L[l] in the lanes l < L, and S_{L/2}[l] = S_L[l] + S_L[l + L/2] -- one
// pass of N / L loads yields L, L/2, ..., L / 2^(levels-1).  A period below 64 cost twice an average pass.
template <typename T, bool MAXABS, typename BF, typename F>
__device__ __forceinline__ void wave_chain_small(const T* __restrict__ xs, int N, int L, int nlev, const PGeom* __restrict__ geom,
                                                 int lane, BF& bf, F&& consume) {
  double tot = wave_fold_small(xs, N, L, lane);
  int q = L;
  for (int lev = 0; lev < nlev; ++lev) {
    const PGeom g = geom[q];
    const double w = (lane < g.nfull) ? g.w_full : g.w_short;
    const double a = lane < q ? (MAXABS ? fabs(tot) : tot * tot * w) : 0.0;
    bf.push(a, q, lane, consume);
    if (lev + 1 < nlev) {
      q >>= 1;
      tot += __shfl(tot, (lane + q) & (kWave - 1), kWave);  // lanes < q read lanes < 2 q: totals of the level above
    }
  }
}

// `queue` (an LDS counter the caller has set to i_first's base + stride, i.e. the first index no wavefront starts
// with): the passes are taken in plan order by whichever wavefront is free instead of strided.
template <typename T, bool LDS, bool MAXABS = false, typename F>
__device__ __forceinline__ void wave_sweep_plan(const T* __restrict__ xs, int N, const PGeom* __restrict__ geom,
                                                const PassPlan* __restrict__ plan, int i_first, int i_end,
                                                int stride, int lane, F&& consume, int* __restrict__ queue = nullptr) {
  Butterfly8<MAXABS> bf;
  bf.reset();
  for (int i = i_first; i < i_end;) {
    int ticket = 0;
    if (queue && lane == 0) ticket = lds_ticket(queue);  // the next pass: claimed now, looked at after this one
    const int p = plan[i].p, m = plan[i].m;
    if (m >= 8) {
      wave_chain_small<T, MAXABS>(xs, N, p, m - 8, geom, lane, bf, consume);
    } else if (m <= 1) {
      bf.push(wave_partial<T, MAXABS, LDS>(xs, N, p, geom[p], lane), p, lane, consume);
    } else if (m == 2) {
      double part[3];
      wave_pass_multi<T, 2, MAXABS, LDS>(xs, p, geom, lane, part);
      bf.push(part[0], p, lane, consume);
      bf.push(part[1], 2 * p, lane, consume);
    } else {
      double part[3];
      wave_pass_multi<T, 4, MAXABS, LDS>(xs, p, geom, lane, part);
      bf.push(part[0], p, lane, consume);
      bf.push(part[1], 2 * p, lane, consume);
      bf.push(part[2], 4 * p, lane, consume);
    }
    i = queue ? lds_ticket_value(ticket) : i + stride;
  }
  bf.flush(lane, consume);
}

// Wavefront argmax of (value, period): largest value, lowest period among equals
// (the reference scans p upward with a strict '>', Periods.py:512).  period 0 = none.
__device__ __forceinline__ void wave_argmax(double& v, int& p) {
  // same pairing as wave_allreduce (l ^ 32, 16, 8, 4, 2, 1), without ds_bpermute; the order "larger value, then
  // lower period, period 0 = none" is total, so both lanes of a pair keep the same winner
  auto take = [](double& v, int& p, double ov, int op) {
    if (op != 0 && (p == 0 || ov > v || (ov == v && op < p))) {
      v = ov;
      p = op;
    }
  };
  {
    unsigned a0 = (unsigned)__double2loint(v), a1 = (unsigned)__double2hiint(v), a2 = (unsigned)p, b0 = a0, b1 = a1, b2 = a2;
    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    const auto r2 = __builtin_amdgcn_permlane32_swap(a2, b2, false, false);
    v = __hiloint2double((int)r1[0], (int)r0[0]);
    p = (int)r2[0];
    take(v, p, __hiloint2double((int)r1[1], (int)r0[1]), (int)r2[1]);
  }
  {
    unsigned a0 = (unsigned)__double2loint(v), a1 = (unsigned)__double2hiint(v), a2 = (unsigned)p, b0 = a0, b1 = a1, b2 = a2;
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    const auto r2 = __builtin_amdgcn_permlane16_swap(a2, b2, false, false);
    v = __hiloint2double((int)r1[0], (int)r0[0]);
    p = (int)r2[0];
    take(v, p, __hiloint2double((int)r1[1], (int)r0[1]), (int)r2[1]);
  }
  take(v, p, dpp_f64<kDppRor8>(v), __builtin_amdgcn_update_dpp(0, p, kDppRor8, 0xF, 0xF, false));
  take(v, p, dpp_f64<kDppHalfMirror>(v), __builtin_amdgcn_update_dpp(0, p, kDppHalfMirror, 0xF, 0xF, false));
  take(v, p, dpp_f64<kDppXor2>(v), __builtin_amdgcn_update_dpp(0, p, kDppXor2, 0xF, 0xF, false));
  take(v, p, dpp_f64<kDppXor1>(v), __builtin_amdgcn_update_dpp(0, p, kDppXor1, 0xF, 0xF, false));
}

// Workgroup-cooperative value of one sweep entry for any flag combination (slow path):
// materialise the projection in `buf`, then norm it.  All threads get the value.
template <typename T>
__device__ __forceinline__ double block_sweep_value(const T* __restrict__ xs, T* __restrict__ buf, int N, int p,
                                                    int gamma_div, unsigned flags, const Tables& tb,
                                                    double* red) {
  project_lds(xs, buf, N, p, flags, tb);
  const double ss = block_sumsq(buf, N, red);
  return periodic_norm_from_sq(ss, N, gamma_div);
}

// Copy one window HBM -> LDS (coalesced, 16-byte vectors when the row is 16-byte aligned).
template <typename T>
__device__ __forceinline__ void zero_pad(T* __restrict__ xs, int N) {
  for (int i = threadIdx.x; i < kPad; i += blockDim.x) xs[N + i] = T(0);
}

template <typename T>
__device__ __forceinline__ void load_window(const T* __restrict__ g, T* __restrict__ xs, int N) {
  constexpr int V = 16 / sizeof(T);
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0 && (N % V) == 0) {
    using vec_t = typename std::conditional<sizeof(T) == 8, double2, float4>::type;
    const vec_t* gv = reinterpret_cast<const vec_t*>(g);
    vec_t* xv = reinterpret_cast<vec_t*>(xs);
    for (int i = threadIdx.x; i < N / V; i += blockDim.x) xv[i] = gv[i];
  } else {
    for (int i = threadIdx.x; i < N; i += blockDim.x) xs[i] = g[i];
  }
}

}  // namespace ph
