// Device-side building blocks of libperiod_hip.so (gfx950 / CDNA4 only).
//
// Everything here works on ONE signal window that is resident in LDS.  A window-projection
// (Periods.project, reference Periods.py:142-219) is a strided fold -> mean -> tile; it has
// no dense contraction, so there is no MFMA anywhere in this library.  The two mappings are
//   * thread-per-residue, rows accumulated in order r = 0..R-1: bit-identical to the
//     reference's np.sum(cp, 0) (Periods.py:194).  Used wherever a projection is
//     materialised (bases, residual updates) and for max|S_p[s]|.
//   * wave-per-period: one 64-lane wavefront owns a candidate period p, lanes own residues,
//     consecutive lanes read consecutive LDS words (conflict-free ds_read_b64), squared
//     sums are combined with wavefront shuffles.  Used by the norm sweeps, where the
//     reference's own norm (BLAS ddot inside np.linalg.norm) has no defined order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ph {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 wavefronts; 4 workgroups of 32 KiB windows fit one CU's 160 KiB LDS
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

// ---------------------------------------------------------------- wavefront / block reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  return v;
}

// Sum over the workgroup; every thread gets the result.  `red` holds >= 2*kMaxWaves doubles.
// Partials are combined in wave order, so the result is identical in all threads.
__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  v = wave_sum(v);
  if ((tid & (kWave - 1)) == 0) red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += red[i];
  __syncthreads();
  return t;
}

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
  double ta = 0.0, tb = 0.0;
  for (int i = 0; i < nw; ++i) {
    ta += red[i];
    tb += red[kMaxWaves + i];
  }
  __syncthreads();
  a = ta;
  b = tb;
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
template <typename T>
__device__ __forceinline__ T column_sum(const T* __restrict__ xs, int j, int p, int n) {
  if (n <= 0) return T(0);
  T s = xs[j];
  for (int r = 1; r < n; ++r) s += xs[r * p + j];
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

// ---------------------------------------------------------------- wave-per-period fold
// ||P_p x||^2 = sum_j S_p[j]^2 / cnt_p[j] for the plain projection.  The two
// count classes are accumulated separately so the hot loop has no division.
// Returns the value in every lane of the calling wavefront.
template <typename T>
__device__ __forceinline__ double wave_proj_sq(const T* __restrict__ xs, int N, int p, int lane) {
  const Fold f(N, p);
  double acc_full = 0.0, acc_short = 0.0;
  for (int j = lane; j < p; j += kWave) {
    const bool full = j < f.nfull;
    const int n = full ? f.rows : f.rows - 1;
    const double s = (double)column_sum(xs, j, p, n);
    if (full)
      acc_full += s * s;
    else
      acc_short += s * s;
  }
  acc_full = wave_sum(acc_full);
  acc_short = wave_sum(acc_short);
  double v = acc_full / (double)f.rows;
  if (f.rows > 1) v += acc_short / (double)(f.rows - 1);
  return v;
}

// max_s |S_p[s]| with row-order sums (Periods.py:327-331), in every lane.
template <typename T>
__device__ __forceinline__ double wave_fold_maxabs(const T* __restrict__ xs, int N, int p, int lane) {
  const Fold f(N, p);
  double best = 0.0;
  for (int j = lane; j < p; j += kWave) {
    const double s = fabs((double)column_sum(xs, j, p, f.count(j)));
    best = fmax(best, s);
  }
  return wave_max(best);
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
