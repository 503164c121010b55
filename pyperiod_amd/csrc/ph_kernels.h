// Kernels of libperiod_hip.so.  One workgroup owns one signal window (or one window and a
// slice of the candidate periods); the window is read from HBM once and lives in LDS for the
// whole algorithm.  Reference citations are file:line into /root/reference/pyPeriod/.
#pragma once

#include <type_traits>

#include "ph_device.h"
#include "ph_pair.h"

namespace ph {

// Dynamic LDS carve-up; every piece starts 16-byte aligned (guide: Guideline 17).
struct Carve {
  unsigned char* base;
  size_t off;
  __device__ explicit Carve(unsigned char* b) : base(b), off(0) {}
  template <typename U>
  __device__ U* take(size_t n) {
    U* p = reinterpret_cast<U*>(base + off);
    off += (n * sizeof(U) + 15) & ~size_t(15);
    return p;
  }
};

__host__ __device__ inline size_t carve_bytes(size_t n, size_t elem) { return (n * elem + 15) & ~size_t(15); }

// The window buffer of workgroup blockIdx.x: LDS normally; for windows longer than the LDS
// (LW == false) a slice of the HBM workspace `gwin` (stride win_stride(len) elements).
__host__ __device__ inline size_t win_stride(size_t len) { return (len + 3) & ~size_t(3); }

template <typename T, bool LW>
__device__ __forceinline__ T* window_buf(Carve& cv, T* gwin, size_t len) {
  if constexpr (LW) {
    return cv.take<T>(len);
  } else {
    return gwin + (size_t)blockIdx.x * win_stride(len);
  }
}

constexpr int kRedDoubles = 2 * kMaxWaves;

// The sweep kernels run 4 workgroups x 8 wavefronts per CU (LDS holds four 32 KiB windows), which
// needs <= 64 VGPRs; measured +15 % over the 5 waves/SIMD the unconstrained allocation gives.
#ifndef PH_STEP1_WAVES
#define PH_STEP1_WAVES 8
#endif

// ======================================================================================
// K1  Periods.project over a batch  (Periods.py:142-219)
//   grid.x = W * chunks; each workgroup projects its window onto p_list[k0 .. k1).
//   Non-orth: means stay in registers and rows stream straight to HBM (write-bound).
//   Orth: projection is materialised in a second LDS buffer, sub-projections removed there.
// ======================================================================================
template <typename T, bool LW>
__global__ __launch_bounds__(kBlock) void k_project_batch(const T* __restrict__ x, int N,
                                                          const int* __restrict__ p_list, int n_p, int chunks,
                                                          unsigned flags, Tables tb, int scratch_len,
                                                          T* __restrict__ gbuf, T* gwin, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  T* xs = window_buf<T, LW>(cv, gwin, N);
  // orth: the whole projection is materialised here; otherwise only one period of means.
  // gbuf != nullptr: the window is too long for two LDS buffers, the second one lives in HBM.
  T* buf = gbuf ? gbuf + (int64_t)blockIdx.x * scratch_len : cv.take<T>(scratch_len);

  const int64_t w = blockIdx.x / chunks;
  const int c = blockIdx.x % chunks;
  const int per = (n_p + chunks - 1) / chunks;
  const int k0 = c * per;
  const int k1 = min(n_p, k0 + per);
  load_window(x + w * (int64_t)N, xs, N);
  __syncthreads();

  const bool trunc = flags & kTrunc;
  const bool single = flags & kSingle;
  constexpr int V = 16 / sizeof(T);
  for (int k = k0; k < k1; ++k) {
    const int p = p_list[k];
    T* orow = out + (w * n_p + k) * (int64_t)N;
    const int lim = single ? min(p, N) : N;
    if (!(flags & kOrth)) {
      // means in row order (bit-identical to the reference), one per residue that exists
      const Fold f(N, p);
      const int np = min(p, N);
      for (int j = threadIdx.x; j < np; j += blockDim.x) buf[j] = residue_mean(xs, f, j, trunc);
      __syncthreads();
      // tile (Periods.py:196-198): coalesced 16-byte stores, n mod p kept incrementally
      if ((N % V) == 0 && (lim % V) == 0) {
        using vec_t = typename std::conditional<sizeof(T) == 8, double2, float4>::type;
        const int stride = blockDim.x * V;
        int j = (threadIdx.x * V) % p;
        const int step = stride % p;
        for (int n = threadIdx.x * V; n < lim; n += stride) {
          T v[V];
          int jj = j;
#pragma unroll
          for (int e = 0; e < V; ++e) {
            v[e] = buf[jj];
            jj = (jj + 1 == p) ? 0 : jj + 1;
          }
          *reinterpret_cast<vec_t*>(orow + n) = *reinterpret_cast<const vec_t*>(v);
          j += step;
          if (j >= p) j -= p;
        }
      } else {
        int j = threadIdx.x % p;
        const int step = blockDim.x % p;
        for (int n = threadIdx.x; n < lim; n += blockDim.x) {
          orow[n] = buf[j];
          j += step;
          if (j >= p) j -= p;
        }
      }
      __syncthreads();  // buf is rewritten by the next period
    } else {
      project_lds(xs, buf, N, p, flags, tb);
      for (int n = threadIdx.x; n < lim; n += blockDim.x) orow[n] = buf[n];
      __syncthreads();
    }
  }
}

// ======================================================================================
// K2  fused all-p sweep  (inner loops Periods.py:501-510, :324-331)
//   grid.x = W * chunks.  Fast path (no flags): wave-per-period, no workgroup barrier after
//   the window load.  Flagged path: workgroup-cooperative full projection per period.
// ======================================================================================
template <typename T, bool LW>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PH_STEP1_WAVES, 8))) void k_sweep(const T* __restrict__ x, int N, int p_lo, int p_hi, int mode,
                                                      int chunks, unsigned flags, Tables tb,
                                                      const PGeom* __restrict__ geom,
                                                      const PassPlan* __restrict__ plan, int n_pass,
                                                      T* __restrict__ gbuf, T* gwin,
                                                      double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  T* xs = window_buf<T, LW>(cv, gwin, N + kPad);
  const bool general = (flags & (kTrunc | kOrth)) && mode != 2;
  T* buf = !general ? nullptr : gbuf ? gbuf + (int64_t)blockIdx.x * N : cv.take<T>(N);
  double* red = cv.take<double>(kRedDoubles);
  int* qctr = cv.take<int>(4);  // pass queue of the norm modes

  const int64_t w = blockIdx.x / chunks;
  const int c = blockIdx.x % chunks;
  const int P = p_hi - p_lo + 1;
  const int per = (P + chunks - 1) / chunks;
  const int pa = p_lo + c * per;
  const int pb = min(p_hi, pa + per - 1);
  load_window(x + w * (int64_t)N, xs, N);
  zero_pad(xs, N);
  if (threadIdx.x == 0) qctr[0] = c * ((n_pass + chunks - 1) / chunks) + (int)(blockDim.x >> 6);
  __syncthreads();
  double* orow = out + w * (int64_t)P;

  if (!general) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    if (mode == 2) {
      wave_sweep<T, true, LW>(xs, N, geom, pa + wv, pb, nw, lane, [&](double v, int p) {
        if ((lane & 7) == 0) orow[p - p_lo] = v;
      });
    } else {
      // norm modes: the workgroups of a window split the pass plan, not the period range
      const int pper = (n_pass + chunks - 1) / chunks;
      const int i0 = c * pper, i1 = min(n_pass, i0 + pper);
      wave_sweep_plan<T, LW>(
          xs, N, geom, plan, i0 + wv, i1, nw, lane,
          [&](double v, int p) {
            if ((lane & 7) == 0) orow[p - p_lo] = periodic_norm_from_sq(v, N, mode == 1 ? p : 0);
          },
          qctr);
    }
  } else {
    for (int p = pa; p <= pb; ++p) {
      const double v = block_sweep_value(xs, buf, N, p, mode == 1 ? p : 0, flags, tb, red);
      if (threadIdx.x == 0) orow[p - p_lo] = v;
    }
  }
}

constexpr int kPairSmallP = 256;  // winners up to this period: means through LDS, subtraction by all threads
constexpr int kPairSplitW = 512;  // threads that share the rows of a short winner's residues (split_row_means)

// ======================================================================================
// m_best step 1  (Periods.py:494-537): repeat { all-p sweep, argmax, subtract } until `num`
// distinct periods are found.  One workgroup per window, one launch per window batch.
// ======================================================================================
template <typename T, bool LW>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PH_STEP1_WAVES, 8))) void k_mbest_step1(const T* __restrict__ x, int N, int num, int p_lo,
                                                        int p_hi, int gamma, unsigned flags, Tables tb,
                                                        const PGeom* __restrict__ geom,
                                                        const PassPlan* __restrict__ plan, int n_pass,
                                                        T* __restrict__ gbuf, T* gwin,
                                                        int max_iters, uint32_t* __restrict__ periods_out,
                                                        double* __restrict__ norms_out, T* __restrict__ rows_out,
                                                        int row_stride, double* __restrict__ dnorm_out,
                                                        int* __restrict__ status_out,
                                                        int* __restrict__ sweeps_out, int small_means) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  T* work = window_buf<T, LW>(cv, gwin, N + kPad);
  const bool general = flags & (kTrunc | kOrth);
  T* buf = !general ? nullptr : gbuf ? gbuf + (int64_t)blockIdx.x * N : cv.take<T>(N);
  double* red = cv.take<double>(kRedDoubles);
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  double* norms = cv.take<double>(num);
  uint32_t* periods = cv.take<uint32_t>(num);
  const int P = p_hi - p_lo + 1;
  uint32_t* skip = cv.take<uint32_t>((P + 31) / 32);
  // means / partial sums of a short winning period (split_row_means); absent when the LDS has no room for them
  T* msm = small_means ? cv.take<T>(kPairSmallP) : nullptr;
  T* prt = small_means ? cv.take<T>(kPairSplitW) : nullptr;

  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  // Basis rows leave step 1 in COMPACT form: row k of the window is the mean vector of its period (the first
  // p elements of the tiled projection; the projection is p-periodic in every flag mode), row_stride elements
  // apart.  Step 2 refines them and writes the (num, N) matrix once, in final order.
  T* rows = rows_out + w * (int64_t)num * row_stride;

  load_window(x + w * (int64_t)N, work, N);
  zero_pad(work, N);
  for (int k = tid; k < num; k += blockDim.x) {
    norms[k] = 0.0;
    periods[k] = 0u;
  }
  for (int k = tid; k < (P + 31) / 32; k += blockDim.x) skip[k] = 0u;
  __syncthreads();
  const double data_norm = uniform_f64(periodic_norm_from_sq(block_sumsq(work, N, red), N, 0));

  int filled = 0;   // `i` of Periods.py:494
  int repeats = 0;  // `iters`
  int status = 0;
  int iters = 0;
  while (filled < num) {
    if (++iters > max_iters) {
      status = 2;
      break;
    }
    // ---- sweep (Periods.py:501-515): strict '>' keeps the lowest p among equal norms
    double best = 0.0;
    int bestp = 0;
    if (!general) {
      // Periods arrive out of order from the pass plan.  The reference compares the rounded norms
      // sqrt(ss)/sqrt(N)[/sqrt(p)] and keeps the lowest period among equal ones (Periods.py:512);
      // the norm is monotone in ss (ss / p in gamma mode), so the sums of squares are compared
      // directly and the square roots and divisions are evaluated only when two candidates are
      // within rounding distance of each other -- and once per lane at the end.
      double best_ss = 0.0;
      wave_sweep_plan<T, LW>(work, N, geom, plan, wv, n_pass, nw, lane, [&](double ss, int p) {
        const bool skipped = (skip[(p - p_lo) >> 5] >> ((p - p_lo) & 31)) & 1u;
        if (skipped || !(ss > 0.0)) return;
        bool take = bestp == 0;
        if (!take) {
          const double lhs = gamma ? ss * (double)bestp : ss;
          const double rhs = gamma ? best_ss * (double)p : best_ss;
          if (lhs > rhs * (1.0 + 1e-14)) {
            take = true;
          } else if (lhs >= rhs * (1.0 - 1e-14)) {
            const double vn = periodic_norm_from_sq(ss, N, gamma ? p : 0);
            const double vb = periodic_norm_from_sq(best_ss, N, gamma ? bestp : 0);
            take = vn > vb || (vn == vb && p < bestp);
          }
        }
        if (take) {
          best_ss = ss;
          bestp = p;
        }
      });
      best = bestp != 0 ? periodic_norm_from_sq(best_ss, N, gamma ? bestp : 0) : 0.0;
      if (!(best > 0.0)) bestp = 0;  // the reference needs p_norm > 0 (Periods.py:497,512)
      wave_argmax(best, bestp);
      if (lane == 0) {
        wbest[wv] = best;
        wbestp[wv] = bestp;
      }
      __syncthreads();
      red_argmax(wbest, wbestp, nw, best, bestp);
      __syncthreads();
    } else {
      for (int p = p_lo; p <= p_hi; ++p) {
        const double v = block_sweep_value(work, buf, N, p, gamma ? p : 0, flags, tb, red);
        const bool skipped = (skip[(p - p_lo) >> 5] >> ((p - p_lo) & 31)) & 1u;
        if (v > best && !skipped) {
          best = v;
          bestp = p;
        }
      }
    }
    if (bestp == 0) {  // reference: max_base is None -> TypeError at Periods.py:520/537
      status = 1;
      break;
    }
    // ---- bookkeeping (Periods.py:518-535); identical in every thread
    int row = -1;
    for (int k = 0; k < num; ++k)
      if (periods[k] == (uint32_t)bestp) row = k;
    int action;  // 0 = subtract only, 1 = store new row, 2 = accumulate into existing row
    __syncthreads();
    if (row >= 0 && repeats < 10) {
      action = 2;
      if (tid == 0) norms[row] += best;
      repeats += 1;
    } else if (row >= 0) {
      action = 0;
      if (tid == 0) skip[(bestp - p_lo) >> 5] |= 1u << ((bestp - p_lo) & 31);
      repeats = 0;
    } else {
      action = 1;
      row = filled;
      if (tid == 0) {
        periods[row] = (uint32_t)bestp;
        norms[row] = best;
      }
      filled += 1;
      repeats = 0;
    }
    // ---- project the winner, update bases row, subtract from the residual (:531-537)
    T* brow = rows + (int64_t)(row < 0 ? 0 : row) * row_stride;
    if (!general && msm && bestp <= kPairSmallP) {
      // a short period: its few residues have hundreds of rows each -- means through LDS (split_row_means, as
      // k_mbest_step1_pair), the subtraction is spread over the workgroup
      split_row_means(work, msm, prt, N, bestp, tid, min((int)blockDim.x, kPairSplitW));
      if (tid < bestp) {
        const T m = msm[tid];
        if (action == 1)
          brow[tid] = m;
        else if (action == 2)
          brow[tid] += m;
      }
      int idx = tid % bestp;
      const int step = blockDim.x % bestp;
      for (int n = tid; n < N; n += blockDim.x) {
        work[n] -= msm[idx];
        idx += step;
        idx = idx >= bestp ? idx - bestp : idx;
      }
    } else if (!general) {
      const Fold f(N, bestp);
      for (int j = tid; j < bestp; j += blockDim.x) {
        const T m = residue_mean(work, f, j, false);
        const int cnt = f.count(j);
        if (action == 1)
          brow[j] = m;
        else if (action == 2)
          brow[j] += m;  // every element of the tiled row would get the same sum (Periods.py:519)
        for (int r = 0; r < cnt; ++r) work[r * bestp + j] -= m;
      }
    } else {
      project_lds(work, buf, N, bestp, flags, tb);
      for (int n = tid; n < N; n += blockDim.x) {
        const T m = buf[n];
        if (n < bestp) {
          if (action == 1)
            brow[n] = m;
          else if (action == 2)
            brow[n] += m;
        }
        work[n] -= m;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  // rows the algorithm never filled keep period 0: step 2 writes them as zeros (np.zeros((num, N)), Periods.py:490)
  for (int k = tid; k < num; k += blockDim.x) {
    periods_out[w * num + k] = periods[k];
    norms_out[w * num + k] = norms[k];
  }
  if (tid == 0) {
    dnorm_out[w] = data_norm;
    status_out[w] = status;
    if (sweeps_out) sweeps_out[w] = status == 2 ? iters - 1 : iters;  // all-p sweeps performed
  }
}

// ======================================================================================
// m_best step 1, window-pair screen (plain projection, fp64 windows that fit the LDS twice).
//   One workgroup owns TWO windows.  LDS holds the pair window pw (element n = {fl32(a[n] sa), fl32(b[n] sb)},
//   ph_pair.h) and ONE fp64 staging buffer; the fp64 residuals live in an HBM workspace (gres; the input itself
//   before the first subtraction).  An iteration of Periods.py:496-537 is
//     1. screen: the pass plan over pw -- every ds_read_b64 / v_pk_add_f32 / address instruction serves both
//        windows; the 1364 x 2 float values land in the (idle) staging buffer;
//     2. per window: best lower bound L = max_q (v_q - rad_q), survivors { q : v_q + rad_q >= L } (pair_radius is a
//        rigorous bound on |screen - exact|, so the exact argmax -- and every period that ties with it in the
//        rounded norm -- is among them; ~1 survivor per sweep);
//     3. per window: residual -> staging, survivors re-evaluated in fp64 and compared exactly as k_mbest_step1
//        does (lazy rounded norms, lowest period among equals), bookkeeping, row-order projection of the winner,
//        and ONE fused pass that subtracts, sums the squares of the new residual, writes it back to the
//        workspace and writes its float image into pw.
//   A window that is not finite / all zero, or whose survivor list overflows, takes every period through the
//   exact evaluation instead (same decisions as k_mbest_step1, slower).
// ======================================================================================
#ifndef PH_PAIR_QUEUE
#define PH_PAIR_QUEUE 1
#endif
constexpr int kPairListCap = 96;
constexpr int kPairCoop = 6;  // up to this many survivors are evaluated by the whole workgroup, one after the other

__device__ __forceinline__ bool pair_usable(double rsq) { return rsq > 0.0 && rsq < 1.79e308; }

// power of two that brings the RMS of a window into [0.7, 1.5)
__device__ __forceinline__ double pair_pick_scale(double rsq, int N) {
  if (!pair_usable(rsq)) return 1.0;
  int e;
  (void)frexp(rsq / (double)N, &e);
  return ldexp(1.0, -(e >> 1));
}

// upper bound of ceil(N / q) without an integer division (the radius only has to be an upper bound)
__device__ __forceinline__ int pair_rows_upper(float fn, int q) {
  return (int)(fn * __builtin_amdgcn_rcpf((float)q) * 1.000001f) + 1;
}

// sum_j S_p[j]^2 / cnt_p[j] of the fp64 window `xs` (LDS), residues j = first, first + stride, ... (p >= 64 path)
__device__ __forceinline__ double pair_exact_part(const double* __restrict__ xs, int p, const PGeom& g, int first, int stride) {
  double part = 0.0;
  for (int j = first; j < p; j += stride) {
    const bool full = j < g.nfull;
    const double s = column_sum(xs, j, p, full ? g.rows : g.rows - 1);
    part = fma(s * s, full ? g.w_full : g.w_short, part);
  }
  return part;
}

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PH_STEP1_WAVES, 8))) void k_mbest_step1_pair(
    const double* __restrict__ x, int W, int N, int num, int p_lo, int p_hi, int gamma,
    const PGeom* __restrict__ geom, const PGeomF* __restrict__ geomf, const PassPlan* __restrict__ plan, int n_pass,
    double* __restrict__ gres, int max_iters, uint32_t* __restrict__ periods_out, double* __restrict__ norms_out,
    double* __restrict__ rows_out, int row_stride, double* __restrict__ dnorm_out, int* __restrict__ status_out,
    int* __restrict__ sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  f2* pw = cv.take<f2>(N + kPad);
  double* stg = cv.take<double>(N + kPad);
  double* red = cv.take<double>(kRedDoubles);
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  double* norms = cv.take<double>(2 * num);
  uint32_t* periods = cv.take<uint32_t>(2 * num);
  const int P = p_hi - p_lo + 1;
  const int SK = (P + 31) / 32;
  uint32_t* skip = cv.take<uint32_t>(2 * SK);
  int* list = cv.take<int>(2 * kPairListCap);
  // per window w: ctl[w] survivors listed, [2+w] filled (`i` of Periods.py:494), [4+w] repeats (`iters`),
  // [6+w] status, [8+w] sweeps done, [10+w] still running, [12+w] every period exactly
  int* ctl = cv.take<int>(16);
  double* dst2 = cv.take<double>(4);  // [w] sum of squares of the scaled residual (unit of the radius), [2+w] its scale
  double* msm = cv.take<double>(kPairSmallP);  // means of a short winning period
  double* prt = cv.take<double>(kPairSplitW);  // partial sums of split_row_means

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const size_t gstride = win_stride((size_t)N);
  float* pwf = reinterpret_cast<float*>(pw);
  const double sqrtN = uniform_f64(sqrt((double)N));

  for (int k = tid; k < 2 * num; k += blockDim.x) {
    norms[k] = 0.0;
    periods[k] = 0u;
  }
  for (int k = tid; k < 2 * SK; k += blockDim.x) skip[k] = 0u;
  if (tid < 16) ctl[tid] = tid == 14 ? nw : 0;  // [14]: pass queue of the screen
  zero_pad(stg, N);
  for (int i = tid; i < kPad; i += blockDim.x) pw[N + i] = f2_zero();
  for (int w = 0; w < 2; ++w) {
    const int64_t gw = 2 * (int64_t)blockIdx.x + w;
    const bool exists = gw < W;
    __syncthreads();
    if (exists) {
      load_window(x + gw * (int64_t)N, stg, N);
    } else {
      for (int n = tid; n < N; n += blockDim.x) stg[n] = 0.0;
    }
    __syncthreads();
    const double rsq = block_sumsq(stg, N, red);
    const double sc = uniform_f64(pair_pick_scale(rsq, N));
    for (int n = tid; n < N; n += blockDim.x) pwf[2 * n + w] = (float)(stg[n] * sc);
    if (tid == 0) {
      dst2[w] = rsq * sc * sc * (1.0 + 1e-9);
      dst2[2 + w] = sc;
      ctl[10 + w] = exists ? 1 : 0;
      ctl[12 + w] = pair_usable(rsq) ? 0 : 1;
      if (exists) dnorm_out[gw] = periodic_norm_from_sq_n(rsq, sqrtN, 0);
    }
  }
#ifdef PH_PAIR_TIMERS
  long long ts[5] = {0, 0, 0, 0, 0};
  long long ts0 = wall_clock64();
  int nsurv_tot = 0, nall = 0;
#define PH_PAIR_MARK(k)                     \
  {                                         \
    const long long now_ = wall_clock64();  \
    ts[k] += now_ - ts0;                    \
    ts0 = now_;                             \
  }
#else
#define PH_PAIR_MARK(k)
#endif
  const float fn = (float)N;
#ifdef PH_CLOCKS
  const long long ck0 = clock64(), wk0 = wall_clock64();
#endif
  for (;;) {
    __syncthreads();
    // (values that are identical in all lanes go to scalar registers as they are read: as vector registers they were
    // spilled around the screen, and every reload is a scratch access through L2 on the critical path of the exact phases)
    const bool act0 = __builtin_amdgcn_readfirstlane(ctl[10]) != 0, act1 = __builtin_amdgcn_readfirstlane(ctl[11]) != 0;
    if (!act0 && !act1) break;
    prio_by_progress(__builtin_amdgcn_readfirstlane(max(ctl[8], ctl[9])));  // sweeps done: keeps the two workgroups of a CU in step (ph_device.h)
    if (wv == 0 && pair_lane() == 0) ctl[0] = ctl[1] = 0;  // (read as `ncand` before the last barrier of the previous round)
    // ---- 1. screen of both windows (Periods.py:501-515 in float); values into the idle staging buffer
    f2* vals = reinterpret_cast<f2*>(stg);
    pair_sweep_plan(
        pw, N, geomf, plan, wv, n_pass, nw, [&](float z, int q) { pair_store1(vals, z, q, p_lo); },
        [&](float z, int q_a, int q_b) { pair_store2(vals, z, q_a, q_b, p_lo); },
        PH_PAIR_QUEUE ? &ctl[14] : nullptr);
    __syncthreads();
    PH_PAIR_MARK(0)
    prio_short_phase();
    {  // phases 2 and 3 of this round
    // The thread and lane indices are RECOMPUTED here (wave number in a scalar register + v_mbcnt): kept live across the
    // screen they were spilled, and the exact phases reloaded them from scratch -- through L2 -- fifteen times per round.
    const int tid = (wv << 6) + pair_lane();
    const int lane = tid & (kWave - 1);
    if (tid == 0) ctl[14] = nw;  // the pass queue of the next screen (barriers in between)
    // ---- 2. survivors of both windows: two passes over the values (a thread sees the same <= 2 entries twice)
    {
      const bool scr0 = act0 && !__builtin_amdgcn_readfirstlane(ctl[12]), scr1 = act1 && !__builtin_amdgcn_readfirstlane(ctl[13]);
      if (scr0 || scr1) {
        const double unit0 = uniform_f64(dst2[0]), unit1 = uniform_f64(dst2[1]);
        double lo0 = -1.0 / 0.0, lo1 = -1.0 / 0.0;
        for (int idx = tid; idx < P; idx += blockDim.x) {
          const int q = p_lo + idx;
          const f2 v = vals[idx];
          const double rad = pair_radius(pair_rows_upper(fn, q), q);
          const uint32_t bit = 1u << (idx & 31);
          double a = (double)v.x - rad * unit0, b = (double)v.y - rad * unit1;
          if (gamma) {
            const double rq = 1.0 / (double)q;
            a *= rq;
            b *= rq;
          }
          if (!(skip[idx >> 5] & bit) && a > lo0) lo0 = a;
          if (!(skip[SK + (idx >> 5)] & bit) && b > lo1) lo1 = b;
        }
        lo0 = wave_max(lo0);
        lo1 = wave_max(lo1);
        if (lane == 0) {
          red[wv] = lo0;
          red[kMaxWaves + wv] = lo1;
        }
        __syncthreads();
        lo0 = uniform_f64(red_combine<true>(red, nw));
        lo1 = uniform_f64(red_combine<true>(red + kMaxWaves, nw));
        // periods that tie with the winner in the ROUNDED norm survive too
        const double thr0 = lo0 - fabs(lo0) * 1e-9, thr1 = lo1 - fabs(lo1) * 1e-9;
        for (int idx = tid; idx < P; idx += blockDim.x) {
          const int q = p_lo + idx;
          const f2 v = vals[idx];
          const double rad = pair_radius(pair_rows_upper(fn, q), q);
          const uint32_t bit = 1u << (idx & 31);
          double a = (double)v.x + rad * unit0, b = (double)v.y + rad * unit1;
          if (gamma) {
            const double rq = 1.0 / (double)q;
            a *= rq;
            b *= rq;
          }
          if (scr0 && !(skip[idx >> 5] & bit) && a >= thr0) {
            const int k = atomicAdd(&ctl[0], 1);
            if (k < kPairListCap) list[k] = q;
          }
          if (scr1 && !(skip[SK + (idx >> 5)] & bit) && b >= thr1) {
            const int k = atomicAdd(&ctl[1], 1);
            if (k < kPairListCap) list[kPairListCap + k] = q;
          }
        }
      }
    }
    __syncthreads();
    PH_PAIR_MARK(1)
    // ---- 3. exact phase, one window at a time through the staging buffer
    for (int w = 0; w < 2; ++w) {
      if (!(w ? act1 : act0)) continue;
      const int64_t gw = 2 * (int64_t)blockIdx.x + w;
      int filled = __builtin_amdgcn_readfirstlane(ctl[2 + w]), repeats = __builtin_amdgcn_readfirstlane(ctl[4 + w]), status = 0,
          iters = __builtin_amdgcn_readfirstlane(ctl[8 + w]);
      const int nlisted = __builtin_amdgcn_readfirstlane(ctl[w]);
      const bool exact_all = __builtin_amdgcn_readfirstlane(ctl[12 + w]) != 0 || nlisted > kPairListCap;
      const int ncand = exact_all ? P : nlisted;
      double* nrm = norms + w * num;
      uint32_t* per = periods + w * num;
      uint32_t* sk = skip + w * SK;
      const int* lst = list + w * kPairListCap;
      __syncthreads();
      if (iters + 1 > max_iters) {
        status = 2;
      } else {
        const double* src = iters == 0 ? x + gw * (int64_t)N : gres + gw * gstride;
        iters += 1;
        load_window(src, stg, N);
        __syncthreads();
        PH_PAIR_MARK(2)
#ifdef PH_PAIR_TIMERS
        nsurv_tot += ncand;
        nall += exact_all ? 1 : 0;
#endif
        double best_ss = 0.0;
        int bestp = 0;
        // same comparison as k_mbest_step1: rounded norms only when two candidates nearly tie
        auto consider = [&](double ss, int p) {
          if (!(ss > 0.0)) return;
          bool take = bestp == 0;
          if (!take) {
            const double lhs = gamma ? ss * (double)bestp : ss;
            const double rhs = gamma ? best_ss * (double)p : best_ss;
            if (lhs > rhs * (1.0 + 1e-14)) {
              take = true;
            } else if (lhs >= rhs * (1.0 - 1e-14)) {
              const double vn = periodic_norm_from_sq_n(ss, sqrtN, gamma ? p : 0);
              const double vb = periodic_norm_from_sq_n(best_ss, sqrtN, gamma ? bestp : 0);
              take = vn > vb || (vn == vb && p < bestp);
            }
          }
          if (take) {
            best_ss = ss;
            bestp = p;
          }
        };
        double best = 0.0;
        if (ncand <= kPairCoop) {
          // few survivors (the normal case): the whole workgroup folds one candidate at a time, a wavefront per
          // 64 residues; partial sums are combined in wave order, so every thread holds the same value
          for (int k = 0; k < ncand; ++k) {
            const int p = exact_all ? p_lo + k : lst[k];
            if ((sk[(p - p_lo) >> 5] >> ((p - p_lo) & 31)) & 1u) continue;
            const PGeom g = geom[p];
            double part = 0.0;
            if (p < 64) {
              if (wv == 0) part = wave_partial_small(stg, N, p, g, lane);
            } else {
              part = pair_exact_part(stg, p, g, tid, blockDim.x);
            }
            consider(block_sum(part, red), p);
          }
          best = bestp != 0 ? periodic_norm_from_sq_n(best_ss, sqrtN, gamma ? bestp : 0) : 0.0;
          if (!(best > 0.0)) bestp = 0;  // the reference needs p_norm > 0 (Periods.py:497,512)
        } else {
          for (int k = wv; k < ncand; k += nw) {
            const int p = exact_all ? p_lo + k : lst[k];
            if ((sk[(p - p_lo) >> 5] >> ((p - p_lo) & 31)) & 1u) continue;
            const PGeom g = geom[p];
            const double part = p < 64 ? wave_partial_small(stg, N, p, g, lane) : pair_exact_part(stg, p, g, lane, kWave);
            consider(wave_sum(part), p);
          }
          best = bestp != 0 ? periodic_norm_from_sq_n(best_ss, sqrtN, gamma ? bestp : 0) : 0.0;
          if (!(best > 0.0)) bestp = 0;
          if (lane == 0) {
            wbest[wv] = best;
            wbestp[wv] = bestp;
          }
          __syncthreads();
          red_argmax(wbest, wbestp, nw, best, bestp);
          __syncthreads();
        }
        PH_PAIR_MARK(3)
        if (bestp == 0) {  // reference: max_base is None -> TypeError at Periods.py:520/537
          status = 1;
        } else {
          // bookkeeping (Periods.py:518-535); identical in every thread
          int row = -1;
          for (int k = 0; k < num; ++k)
            if (per[k] == (uint32_t)bestp) row = k;
          int action;  // 0 = subtract only, 1 = store new row, 2 = accumulate into existing row
          __syncthreads();
          if (row >= 0 && repeats < 10) {
            action = 2;
            if (tid == 0) nrm[row] += best;
            repeats += 1;
          } else if (row >= 0) {
            action = 0;
            if (tid == 0) sk[(bestp - p_lo) >> 5] |= 1u << ((bestp - p_lo) & 31);
            repeats = 0;
          } else {
            action = 1;
            row = filled;
            if (tid == 0) {
              per[row] = (uint32_t)bestp;
              nrm[row] = best;
            }
            filled += 1;
            repeats = 0;
          }
          // project the winner (row order), update the compact basis row, subtract (:531-537).  The new residual is
          // not needed in LDS again: it goes straight to the workspace and, as floats, into pw.  The scale of the
          // float image only has to keep the RMS near 1; it is renewed when the residual has shrunk by 2^16.
          double* brow = rows_out + (gw * (int64_t)num + (row < 0 ? 0 : row)) * row_stride;
          const bool more = filled < num;
          const double sc = uniform_f64(dst2[2 + w]);
          double* dst = gres + gw * gstride;
          double acc = 0.0;
          const Fold f(N, bestp);
          if (bestp <= kPairSmallP) {
            // a short period: its few residues have hundreds of rows each -- means through LDS, the subtraction is spread
            // over the workgroup
            // (the usual winner of m_best_gamma); split_row_means deals the rows of a residue to several threads
            split_row_means(stg, msm, prt, N, bestp, tid, kPairSplitW);
            if (tid < bestp) {
              const double m = msm[tid];
              if (action == 1)
                brow[tid] = m;
              else if (action == 2)
                brow[tid] += m;
            }
            if (more) {
              int idx = tid % bestp;
              const int step = blockDim.x % bestp;
              for (int n = tid; n < N; n += blockDim.x) {
                const double v = stg[n] - msm[idx];
                dst[n] = v;
                pwf[2 * n + w] = (float)(v * sc);
                acc = fma(v, v, acc);
                idx += step;
                idx = idx >= bestp ? idx - bestp : idx;
              }
            }
          } else {
            for (int j = tid; j < bestp; j += blockDim.x) {
              const double m = residue_mean(stg, f, j, false);
              const int cnt = f.count(j);
              if (action == 1)
                brow[j] = m;
              else if (action == 2)
                brow[j] += m;
              if (more) {
                for (int r = 0; r < cnt; ++r) {
                  const int n = r * bestp + j;
                  const double v = stg[n] - m;
                  dst[n] = v;
                  pwf[2 * n + w] = (float)(v * sc);
                  acc = fma(v, v, acc);
                }
              }
            }
          }
          if (more) {
            const double rsq = block_sum(acc, red);
            const double unit = rsq * sc * sc;
            const bool ok = pair_usable(rsq);
            if (ok && unit < 9.0e-13 * (double)N) {  // RMS of the float image below 2^-20: renew the scale
              const double sc2 = uniform_f64(pair_pick_scale(rsq, N));
              const float up = (float)(sc2 / sc);  // a power of two: the rescaled image is the image at the new scale
              __syncthreads();
              for (int n = tid; n < N; n += blockDim.x) pwf[2 * n + w] *= up;
              if (tid == 0) {
                dst2[w] = rsq * sc2 * sc2 * (1.0 + 1e-9);
                dst2[2 + w] = sc2;
              }
            } else if (tid == 0) {
              dst2[w] = unit * (1.0 + 1e-9);
            }
            if (tid == 0) ctl[12 + w] = ok ? 0 : 1;
          }
        }
      }
      __syncthreads();
      if (tid == 0) {
        ctl[2 + w] = filled;
        ctl[4 + w] = repeats;
        ctl[6 + w] = status;
        ctl[8 + w] = iters;
        ctl[10 + w] = (status == 0 && filled < num) ? 1 : 0;
      }
      PH_PAIR_MARK(4)
    }
    }  // phases 2 and 3
  }
#ifdef PH_CLOCKS
  if ((blockIdx.x % 16) == 7 && tid == 0) {
    const long long ck2 = clock64(), wk2 = wall_clock64();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    printf("PAIRWG %d start %lld end %lld cycles %lld MHz %.0f hwid %x xcc %x\n", (int)blockIdx.x, wk0, wk2, ck2 - ck0,
           100.0 * (double)(ck2 - ck0) / (double)(wk2 - wk0), hwid, xcc & 15);
  }
#endif
#ifdef PH_PAIR_TIMERS
  if (blockIdx.x < 6 && tid == 0)
    printf("pair timers (100 MHz ticks) screen %lld scan %lld load %lld exact %lld update %lld  candidates %d exact_all %d sweeps %d %d\n",
           ts[0], ts[1], ts[2], ts[3], ts[4], nsurv_tot, nall, ctl[8], ctl[9]);
#endif
  __syncthreads();
  // rows the algorithm never filled keep period 0: step 2 writes them as zeros (np.zeros((num, N)), Periods.py:490)
  for (int w = 0; w < 2; ++w) {
    const int64_t gw = 2 * (int64_t)blockIdx.x + w;
    if (gw >= W) continue;
    for (int k = tid; k < num; k += blockDim.x) {
      periods_out[gw * num + k] = periods[w * num + k];
      norms_out[gw * num + k] = norms[w * num + k];
    }
    if (tid == 0) {
      status_out[gw] = ctl[6 + w];
      if (sweeps_out) sweeps_out[gw] = ctl[8 + w];
    }
  }
}

__device__ __forceinline__ void ram_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ======================================================================================
// m_best step 2  (Periods.py:540-598): factor refinement, including the reference's
// quirks: stale `p` in gamma mode (:559,572), `nq` = norm of the LAST factor's projection
// (:569-572), no advance of i after a split (:581-594).  One workgroup per window.
// norms_io holds raw norms on entry and powers (norms / ||data||, :600) on exit.
//   Compact rows.  Basis row i is the tiled mean vector m of its period p (step 1 writes nothing else, a
//   split keeps it so: the factor's projection is f-periodic, f | p), so step 1 hands over only the first
//   p elements of every row and np.insert (:585-594) becomes a permutation of row slots: a split writes
//   two compact rows (the projection into the slot of the row that falls off the end, the remainder in
//   place) and moves no data.  The (num, N) matrix is written exactly once, in final order, each row as
//   soon as the loop has passed it.
//   For a divisor f of p
//     S_f[j] = sum_{n < N, n = j (mod f)} row[n] = sum_{k < p, k = j (mod f)} cnt_p[k] m[k],
//   so the factor norms (plain projection, fp64) fold the count-scaled p-vector -- a "window" of length p
//   whose every residue mod f has p/f rows -- instead of the N-element row.  The count weights 1/cnt_f[j]
//   are those of the real window: the fold runs on geom[f] with rows := p/f + 1, whose one extra row for
//   the residues j < nfull_f reads zeros stored behind the vector.
// ======================================================================================
// geom[p] computed in registers (same expressions as prepare_geom on the host; IEEE divisions)
__device__ __forceinline__ PGeom make_geom(int N, int p) {
  const int rows = (N + p - 1) / p;
  PGeom g;
  g.rows = rows;
  g.nfull = p - (rows * p - N);
  g.w_full = 1.0 / (double)rows;
  g.w_short = rows > 1 ? 1.0 / (double)(rows - 1) : 0.0;
  return g;
}

constexpr int kStep2Pre = 3;  // a staged row has at most 3 elements per thread (p <= 3 * blockDim)
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;

// rowbuf[n] = row[n mod p], n < N: the tiled row of a compact vector
template <typename T>
__device__ __forceinline__ void expand_row(const T* __restrict__ src, int p, T* __restrict__ dst, int N) {
  const int step = blockDim.x % p;
  int idx = threadIdx.x % p;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    dst[n] = src[idx];
    idx += step;
    if (idx >= p) idx -= p;
  }
}

template <typename T, bool LW>
__global__ __launch_bounds__(kBlockWide) __attribute__((amdgpu_waves_per_eu(PH_STEP1_WAVES, 8))) void k_mbest_step2(int N, int num, int gamma, int stale_p, unsigned flags,
                                                        Tables tb, const PGeom* __restrict__ geom, int max_fac,
                                                        T* __restrict__ gbuf, T* gwin,
                                                        uint32_t* __restrict__ periods_io,
                                                        double* __restrict__ norms_io, T* __restrict__ bases_out,
                                                        const double* __restrict__ dnorm,
                                                        const int* __restrict__ status, T* __restrict__ rows_io,
                                                        int row_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  T* rowbuf = window_buf<T, LW>(cv, gwin, N + kPad);
  T* buf = gbuf ? gbuf + (int64_t)blockIdx.x * N : cv.take<T>(N);
  double* red = cv.take<double>(kRedDoubles);
  double* norms = cv.take<double>(num);
  uint32_t* periods = cv.take<uint32_t>(num);
  double* fvals = cv.take<double>(max_fac > 0 ? max_fac : 1);
  int* ffac = cv.take<int>(max_fac > 0 ? max_fac : 1);
  PGeom* fgeom = cv.take<PGeom>(kMaxWaves);  // per wavefront: geometry of the fold of a periodic row
  int* fa = cv.take<int>(num);               // divisor list of row k's period: tb.fac_q[fa[k] .. fb[k])
  int* fb = cv.take<int>(num);
  int* slot = cv.take<int>(num);             // compact row of logical row k: rows + slot[k] * row_stride
  int* ffac_next = cv.take<int>(max_fac > 64 ? max_fac : 64);  // divisor list of the staged row

  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const bool general = flags & (kTrunc | kOrth);
  const bool periodic_rows = sizeof(T) == 8 && !general;
  T* rows = rows_io + w * (int64_t)num * row_stride;
  for (int k = tid; k < num; k += blockDim.x) {
    norms[k] = norms_io[w * num + k];
    const uint32_t per = periods_io[w * num + k];
    periods[k] = per;
    fa[k] = tb.fac_off[per];
    fb[k] = tb.fac_off[per + 1];
    slot[k] = k;
  }
  zero_pad(rowbuf, N);
  __syncthreads();

#ifdef PH_STEP2_TIMERS
  long long t2[6] = {0, 0, 0, 0, 0, 0};
  long long t20 = wall_clock64();
  const long long t2start = t20;
  int nrows_done = 0, nsplit = 0;
#define PH_S2_MARK(k)                      \
  {                                        \
    const long long now_ = wall_clock64(); \
    t2[k] += now_ - t20;                   \
    t20 = now_;                            \
  }
#else
#define PH_S2_MARK(k)
#endif
  const int gdiv = gamma ? stale_p : 0;
  int i = (status[w] == 0) ? 0 : num;  // a window whose step 1 failed is passed through
  // a row whose period has no proper divisor (a prime) has nothing to test and is not even read
  auto next_row = [&](int r) {
    while (r < num && fa[r] == fb[r]) ++r;
    return r;
  };
  // The rows are short and known one row ahead: the next row's elements and divisor list are copied
  // HBM -> LDS by LDS-DMA (global_load_lds: no registers, the 64-VGPR fold code has none to spare) into the
  // unused upper part of rowbuf while the current row is folded -- the loop is otherwise a chain of dependent
  // HBM round trips (row, divisor list, geometry) per row.  The barrier behind the folds drains the DMA
  // (hipcc waits vmcnt(0) before __syncthreads).
  const T* stage = nullptr;
  int stage_row = -1;
  // The (num, N) matrix is written exactly once, in final order: logical rows below the loop position can
  // no longer change (a split inserts at i and moves rows >= i only), so they are tiled out as the loop
  // passes them -- the stores drain while the next rows are folded.  Rows step 1 never filled (period 0)
  // are zeros (np.zeros((num, N)), Periods.py:490).
  T* out = bases_out + w * (int64_t)num * N;
  int done = 0;
  auto flush_rows = [&](int upto) {
    for (; done < upto; ++done) {
      const int pr = (int)periods[done];
      T* dst = out + (int64_t)done * N;
      if (pr == 0) {
        for (int n = tid; n < N; n += blockDim.x) dst[n] = T(0);
      } else {
        expand_row(rows + (int64_t)slot[done] * row_stride, pr, dst, N);
      }
    }
  };
  i = next_row(i);
  PH_S2_MARK(0)
  while (i < num) {
    const int per = (int)periods[i];
    const int a = fa[i], b = fb[i];
    const T* src = rows + (int64_t)slot[i] * row_stride;
    // v[k] = cnt_p[k] m[k] for k < per, then zeros for the extra row and the tail reads of the last group
    const int zend = per + (per >> 1) + 256;
    const bool short_row = periodic_rows && zend <= N + kPad;
    if (short_row) {
      const PGeom gp = make_geom(N, per);
      const double cf = (double)gp.rows, cs = (double)(gp.rows - 1);
      if (stage_row == i) {  // the staging area may overlap [0, zend): read, barrier, write
        double t[kStep2Pre];
#pragma unroll
        for (int u = 0; u < kStep2Pre; ++u) {
          const int k = tid + u * blockDim.x;
          t[u] = k < per ? (double)stage[k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kStep2Pre; ++u) {
          const int k = tid + u * blockDim.x;
          if (k < zend) rowbuf[k] = (T)(t[u] * (k < gp.nfull ? cf : cs));
        }
        for (int k = tid + kStep2Pre * blockDim.x; k < zend; k += blockDim.x) rowbuf[k] = T(0);
        if (tid < b - a) ffac[tid] = ffac_next[tid];
      } else {
        for (int k = tid; k < zend; k += blockDim.x)
          rowbuf[k] = (T)(k < per ? (double)src[k] * (k < gp.nfull ? cf : cs) : 0.0);
        if (tid < b - a) ffac[tid] = tb.fac_q[a + tid];
      }
    } else {  // fp32 rows, trunc / orth modes, periods beyond 2N/3: the tiled row itself
      expand_row(src, per, rowbuf, N);
      if (!general && tid < b - a) ffac[tid] = tb.fac_q[a + tid];
    }
    __syncthreads();
    PH_S2_MARK(1)
    const int inext = next_row(i + 1);
    stage_row = -1;
    if (LW && short_row && inext < num) {  // LW == false: rowbuf is an HBM workspace, nothing to stage into
      const int pn = (int)periods[inext];
      const int so = (zend + 1) & ~1;          // 16-byte aligned, behind everything the folds of row i read
      const int npieces = (pn + 127) >> 7;     // 64 lanes x 16 B = 128 doubles per wave instruction
      if (pn <= kStep2Pre * (int)blockDim.x && pn + (pn >> 1) + 256 <= N + kPad && so + 128 * npieces <= N &&
          128 * npieces <= row_stride && fb[inext] - fa[inext] <= 64) {
        const T* srcn = rows + (int64_t)slot[inext] * row_stride;  // 128 * npieces <= row_stride: stays inside the row
        for (int j = wv; j < npieces; j += nw)
          __builtin_amdgcn_global_load_lds((global_void_ptr)(srcn + 128 * j + 2 * lane),
                                           (lds_void_ptr)(rowbuf + so + 128 * j), 16, 0, 0);
        if (wv == nw - 1)  // reads up to 63 words behind the list: the device table has that slack
          __builtin_amdgcn_global_load_lds((global_void_ptr)(tb.fac_q + fa[inext] + lane), (lds_void_ptr)ffac_next, 4, 0, 0);
        stage = rowbuf + so;
        stage_row = inext;
      }
    }
    double top = 0.0, last = 0.0;
    int topf = -1;
    if (!general) {
      // one wavefront per factor: ||P_f row||^2 = sum_j S_f[j]^2 / cnt_f[j]
      for (int k = wv; k < b - a; k += nw) {
        const int f = ffac[k];
        // one call site for both row forms (a second inlined copy of the fold costs spills): the geometry sits
        // in this wavefront's LDS slot.  Compact row: a "window" of length per with per / f rows per residue,
        // weights by the real counts of f (rows + 1: see the header; the f < 64 path ignores `rows`).
        if (lane == 0) {
          PGeom g;
          if (short_row) {
            g = make_geom(N, f);
            g.rows = per / f + 1;
          } else {
            g = geom[f];
          }
          fgeom[wv] = g;
        }
        ram_wave_sync();
        const double ss = wave_sum(wave_partial<T, false, LW>(rowbuf, short_row ? per : N, f, fgeom[wv], lane));
        ram_wave_sync();  // the slot is rewritten for this wavefront's next factor
        if (lane == 0) fvals[k] = periodic_norm_from_sq(ss, N, gdiv);
      }
      PH_S2_MARK(2)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA of the next row has landed
      __syncthreads();
      PH_S2_MARK(3)
      // rows below i are final: their stores drain behind the scan below and the next row's folds (issued
      // here, after the wait above, so that the DMA is not held up behind them)
      flush_rows(i);
      for (int k = 0; k < b - a; ++k) {  // the reference's scan order (Periods.py:549-563)
        const double v = fvals[k];
        if (v > top) {
          top = v;
          topf = ffac[k];
        }
        last = v;
      }
    } else {
      flush_rows(i);
      for (int k = a; k < b; ++k) {
        const int f = tb.fac_q[k];
        const double v = block_sweep_value(rowbuf, buf, N, f, gdiv, flags, tb, red);
        if (v > top) {
          top = v;
          topf = f;
        }
        last = v;
      }
    }
    bool split = false;
    if (topf >= 0) {
      bool present = false;
      for (int k = 0; k < num; ++k) present |= (periods[k] == (uint32_t)topf);
      if (!present) {
        double floor_q = norms[0];
        for (int k = 1; k < num; ++k) floor_q = fmin(floor_q, norms[k]);
        split = (last + top) > (norms[num - 1] + norms[i]) && last > floor_q && top > floor_q;
      }
    }
    const int s_here = slot[i], s_last = slot[num - 1];
    __syncthreads();
    PH_S2_MARK(4)
#ifdef PH_STEP2_TIMERS
    nrows_done += 1;
    nsplit += split ? 1 : 0;
#endif
    if (!split) {
      i = inext;
      continue;
    }
    // ---- split (:581-594): materialise the winning factor's projection exactly;
    //   logical rows i+2.. <- rows i+1.. ; row i+1 <- row i - proj ; row i <- proj ; the last row falls off.
    //   In compact form: proj (topf elements) goes into the slot of the row that falls off, row i - proj
    //   (per elements) replaces row i in place, and the slots are renumbered -- no row moves.
    stage_row = -1;
    if (short_row) {  // the projection wants the tiled row
      expand_row(src, per, rowbuf, N);
      __syncthreads();
    }
    project_lds(rowbuf, buf, N, topf, flags, tb);
    {
      T* dst = rows + (int64_t)s_last * row_stride;  // i == num - 1: the split row itself falls off
      if (i + 1 < num) {
        T* rem = rows + (int64_t)s_here * row_stride;
        for (int k = tid; k < per; k += blockDim.x) rem[k] = rowbuf[k] - buf[k];
      }
      for (int k = tid; k < topf; k += blockDim.x) dst[k] = buf[k];
    }
    if (tid == 0) {
      for (int r = num - 1; r >= i + 2; --r) {
        norms[r] = norms[r - 1];
        periods[r] = periods[r - 1];
        fa[r] = fa[r - 1];
        fb[r] = fb[r - 1];
        slot[r] = slot[r - 1];
      }
      if (i + 1 < num) {
        norms[i + 1] = last;  // the old row keeps its period, weakened (:582-584)
        periods[i + 1] = (uint32_t)per;
        fa[i + 1] = a;
        fb[i + 1] = b;
        slot[i + 1] = s_here;
      }
      norms[i] = top;
      periods[i] = (uint32_t)topf;
      fa[i] = tb.fac_off[topf];
      fb[i] = tb.fac_off[topf + 1];
      slot[i] = s_last;
    }
    __threadfence_block();
    __syncthreads();
    i = next_row(i);  // the new row i is examined next: i is not advanced (:581-594)
  }
  __syncthreads();
  flush_rows(num);
  PH_S2_MARK(5)
#ifdef PH_STEP2_TIMERS
  if ((blockIdx.x % 128) == 5 && tid == 0)
    printf("step2 wg %d (100 MHz ticks): init %lld stage %lld folds %lld dma+barrier %lld flush+scan %lld rest(split,final flush) %lld total %lld rows %d splits %d\n",
           (int)blockIdx.x, t2[0], t2[1], t2[2], t2[3], t2[4], t2[5], wall_clock64() - t2start, nrows_done, nsplit);
#endif
  const double dn = dnorm[w];
  for (int k = tid; k < num; k += blockDim.x) {
    periods_io[w * num + k] = periods[k];
    norms_io[w * num + k] = norms[k] / dn;
  }
}

}  // namespace ph
#include "ph_s2l.h"  // flat exact evaluation (shared) and the window-pair kernel
namespace ph {

// ======================================================================================
// Periods.small_to_large  (Periods.py:246-287).  Sequential in p; one workgroup per window.
//   Plain projection: screen every p with ||r - P r||^2 = ||r||^2 - ||P r||^2 (one LDS pass),
//   and evaluate the reference's expression exactly (second pass) whenever the screen is
//   within its own error bound of the threshold, so the accept decision is the exact one.
// ======================================================================================
// periods screened speculatively per round (8 waves x 4 periods).  Screens beyond an accepted
// period are discarded, (batch - 1) / 2 of them per accept on average: 32 beats 64 (9.4 vs 9.9 ms
// on the config-4 shard) and 16 / 24 / 48.
constexpr int kS2LBatch = 32;

template <typename T, bool LW>
__global__ __launch_bounds__(kBlockWide) __attribute__((amdgpu_waves_per_eu(PH_STEP1_WAVES, 8))) void k_small_to_large(const T* __restrict__ x, int N, double thresh,
                                                           int n_periods, unsigned flags, Tables tb,
                                                           const PGeom* __restrict__ geom, T* __restrict__ gbuf,
                                                           T* gwin, int cap, int* __restrict__ counts,
                                                           int* __restrict__ periods_out,
                                                           double* __restrict__ powers_out,
                                                           T* __restrict__ bases_out, int* __restrict__ status_out,
                                                           int* __restrict__ max_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  T* work = window_buf<T, LW>(cv, gwin, N + kPad);
  const bool general = flags & (kTrunc | kOrth);
  T* buf = !general ? nullptr : gbuf ? gbuf + (int64_t)blockIdx.x * N : cv.take<T>(N);
  double* red = cv.take<double>(kRedDoubles);
  double* psq = cv.take<double>(kS2LBatch);
  int* cand_slot = cv.take<int>(4);
  T* msm = general ? nullptr : cv.take<T>(kBlockWide);  // means of a candidate period <= kBlockWide (s2l_flat_* below)

  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  load_window(x + w * (int64_t)N, work, N);
  zero_pad(work, N);
  __syncthreads();
  double rsq = block_sumsq(work, N, red);
  const double sqrtN = uniform_f64(sqrt((double)N));
  const double dn = uniform_f64(sqrt(rsq) / sqrtN);  // data_norm, Periods.py:269
  double rn = dn;                                    // periodic_norm(residual)
  int count = 0;

#ifdef PH_S2L_TIMERS
  long long ts[4] = {0, 0, 0, 0};
  long long ts0 = wall_clock64();
#define PH_S2L_MARK(k)                      \
  {                                         \
    const long long now_ = wall_clock64();  \
    ts[k] += now_ - ts0;                    \
    ts0 = now_;                             \
  }
#else
#define PH_S2L_MARK(k)
#endif
  int p = 2;
  while (p <= n_periods) {
    int cand = p;  // the period that gets the exact evaluation of Periods.py:274-281
    if (!general) {
      // The residual only changes when a period is accepted, so a run of periods can be
      // screened in parallel (wave-per-period): ||r - P r||^2 = ||r||^2 - ||P r||^2 gives an
      // estimate of the reference's norm drop.  The first period whose estimate, plus a bound
      // on its cancellation error, reaches the threshold is evaluated exactly below; results
      // screened beyond an accepted period are discarded and recomputed.
      const int hi = min(n_periods, p + kS2LBatch - 1);
      wave_sweep<T, false, LW, true>(work, N, geom, p + wv, hi, nw, lane, [&](double ss, int q) {
        if (lane == 0) psq[q - p] = ss;
      });
      __syncthreads();
      if (wv == 0) {  // one lane per screened period; the first flagged one is the candidate
        const int q = p + lane;
        bool flag = false;
        if (q <= hi) {
          // The decision below is taken on t_e = fl(sum (r - m)^2), the screen has t_s = rsq - psq.
          // Both approximate T = ||r||^2 - ||P r||^2:
          //   |psq - ||P r||^2| <= (2 R + q/64 + 16) eps ||r||^2   (S_j: R-term sums, |S_j| A_j / cnt_j <= Q_j
          //                                                         by Cauchy-Schwarz; then a positive sum)
          //   |rsq - ||r||^2|, |t_e - T| <= (N / 512 + 16) eps ||r||^2 each (strided partial sums + tree)
          // so |t_s - t_e| <= D = (1.5 N + 256) eps rsq for every q >= 2 (2 R <= N, q / 64 <= N / 128).
          // float windows: the means and r - m are rounded to float, adding (4 + R^2 eps_f) eps_f rsq.
          // |sqrt(t_s) - sqrt(t_e)| <= min(D / sqrt(t_s), sqrt(D)); the remaining operations of
          // (rn - sqrt(t) / sqrtN) / dn are shared by both paths up to 8 eps (<= 1e-13).
          const double tsq = fmax(rsq - psq[lane], 0.0);
          const double est = (rn - sqrt(tsq) / sqrtN) / dn;
          double kappa = (1.5 * (double)N + 256.0) * 2.220446049250313e-16;
          if (sizeof(T) == 4) {
            const double rf = (double)geom[q].rows * 5.9604644775390625e-08;
            kappa += (4.0 + rf * (double)geom[q].rows) * 5.9604644775390625e-08;
          }
          const double D = kappa * rsq;
          const double dsq = fmin(D / fmax(sqrt(tsq), 1e-300), sqrt(D));
          const double err = dsq / sqrtN / dn + 1e-13;
          flag = !(est + err <= thresh);  // NaN -> evaluate
        }
        const unsigned long long mask = __ballot(flag);
        if (lane == 0) *cand_slot = mask ? p + __ffsll((long long)mask) - 1 : -1;
      }
      __syncthreads();  // also: psq is rewritten by the next round
      PH_S2L_MARK(0)
      cand = *cand_slot;
      if (cand < 0) {
        p = hi + 1;
        continue;
      }
    }
    // exact evaluation (row-order means, direct sum of squares of the trial residual)
    double tsq = 0.0;
    const bool flat = !general && cand <= kBlockWide;
    if (flat) {
      tsq = s2l_flat_trial(work, msm, msm, N, cand, tid, (int)blockDim.x);
    } else if (!general) {
      const Fold f(N, cand);
      for (int j = tid; j < cand; j += blockDim.x) {
        const T m = residue_mean(work, f, j, false);
        const int cnt = f.count(j);
        for (int r = 0; r < cnt; ++r) {
          const double t = (double)(work[r * cand + j] - m);
          tsq += t * t;
        }
      }
    } else {
      project_lds(work, buf, N, cand, flags, tb);
      for (int n = tid; n < N; n += blockDim.x) {
        const double t = (double)(work[n] - buf[n]);
        tsq += t * t;
      }
    }
    tsq = block_sum(tsq, red);
    PH_S2L_MARK(1)
    const double tn = uniform_f64(sqrt(tsq) / sqrtN);
    const double imposed = uniform_f64((rn - tn) / dn);
    if (imposed > thresh) {  // strict, Periods.py:281
      T* brow = (bases_out && count < cap) ? bases_out + (w * cap + count) * (int64_t)N : nullptr;
      if (flat) {
        s2l_flat_update(work, msm, N, cand, tid, (int)blockDim.x, brow, [](int, T) {});
      } else if (!general) {
        const Fold f(N, cand);
        for (int j = tid; j < cand; j += blockDim.x) {
          const T m = residue_mean(work, f, j, false);
          const int cnt = f.count(j);
          for (int r = 0; r < cnt; ++r) {
            const int n = r * cand + j;
            if (brow) brow[n] = m;
            work[n] -= m;
          }
        }
      } else {
        for (int n = tid; n < N; n += blockDim.x) {
          const T m = buf[n];
          if (brow) brow[n] = m;
          work[n] -= m;
        }
      }
      if (tid == 0 && count < cap) {
        periods_out[w * cap + count] = cand;
        powers_out[w * cap + count] = imposed;
      }
      count += 1;
      rsq = tsq;
      rn = tn;
    }
    __syncthreads();
    PH_S2L_MARK(2)
    p = cand + 1;
  }
#ifdef PH_S2L_TIMERS
  if (w < 6 && tid == 0) printf("s2l timers (100 MHz ticks) screen %lld exact %lld update %lld  accepts %d\n", ts[0], ts[1], ts[2], count);
#endif
  if (tid == 0) {
    counts[w] = count;
    status_out[w] = count > cap ? 3 : 0;
    if (count > cap) atomicMax(max_count, count);  // rare: the host retries with this capacity
  }
}

// ======================================================================================
// Periods.best_correlation  (Periods.py:289-349).  One workgroup per window.
// ======================================================================================
constexpr int kCandCap = 256;  // screened candidates per sweep kept for exact evaluation

struct CandCtl {
  unsigned long long lbits;  // running lower bound on the best value (bits of a non-negative double)
  int ncand;
  int nsurv;
};

template <typename T, bool LW>
__global__ __launch_bounds__(kBlockWide) __attribute__((amdgpu_waves_per_eu(PH_STEP1_WAVES, 8))) void k_best_correlation(const T* __restrict__ x, int N, int num,
                                                             int max_length, double ratio, unsigned flags,
                                                             Tables tb, const PGeom* __restrict__ geom,
                                                             const PassPlan* __restrict__ plan, int n_pass,
                                                             T* __restrict__ gbuf, T* gwin,
                                                             uint32_t* __restrict__ periods_out,
                                                             double* __restrict__ norms_out,
                                                             T* __restrict__ bases_out,
                                                             int* __restrict__ status_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  T* work = window_buf<T, LW>(cv, gwin, N + kPad);
  const bool general = flags & (kTrunc | kOrth);
  T* buf = !general ? nullptr : gbuf ? gbuf + (int64_t)blockIdx.x * N : cv.take<T>(N);
  double* red = cv.take<double>(kRedDoubles);
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  int* cand_q = cv.take<int>(kCandCap);
  double* cand_hi = cv.take<double>(kCandCap);
  CandCtl* ctl = cv.take<CandCtl>(1);

  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  T* bases = bases_out + w * (int64_t)num * N;
  load_window(x + w * (int64_t)N, work, N);
  zero_pad(work, N);
  __syncthreads();
  const double sqrtN = sqrt((double)N);
  const double og = sqrt(block_sumsq(work, N, red)) / sqrtN;  // :316
  double old_norm = og;
  int status = 0;

  for (int i = 0; i < num; ++i) {
    T* brow = bases + (int64_t)i * N;
    uint32_t keep_p = 0u;
    double keep_n = 0.0;
    bool zero_row = true;
    if (status == 0) {
      // argmax over (p, s) of |sum(x[s::p])|, strict '>' in p-major order (:324-331).
      // Screen: the multi-period passes of the pass plan give max_s |S_p[s]| for up to three periods
      // per fold, but from class sums, i.e. not in the reference's row order.  |S~ - S| <= R 2^-53
      // sum|x| bounds the difference, so every period whose screened value is within that radius
      // of the best is re-evaluated with the row-order pass and the exact values are compared.
      double asum = 0.0;
      for (int n = tid; n < N; n += blockDim.x) asum += fabs((double)work[n]);
      asum = block_sum(asum, red);
      if (tid == 0) {
        ctl->lbits = 0ull;
        ctl->ncand = 0;
        ctl->nsurv = 0;
      }
      __syncthreads();
      {
        const double radius = 4.0 * 1.1102230246251565e-16 * asum;  // 4 * 2^-53 * sum|x| per row
        double lrun = 0.0;
        wave_sweep_plan<T, LW, true>(work, N, geom, plan, wv, n_pass, nw, lane, [&](double v, int q) {
          const double e = radius * (double)(geom[q].rows + 4);
          const double hi = v + e, lo = v - e;
          lrun = fmax(lrun, __longlong_as_double((long long)*(volatile unsigned long long*)&ctl->lbits));
          if ((lane & 7) == 0 && hi > 0.0 && hi >= lrun) {
            const int idx = atomicAdd(&ctl->ncand, 1);
            if (idx < kCandCap) {
              cand_q[idx] = q;
              cand_hi[idx] = hi;
            }
          }
          if (lo > lrun) {
            lrun = lo;
            if ((lane & 7) == 0) atomicMax(&ctl->lbits, (unsigned long long)__double_as_longlong(lo));
          }
        });
      }
      __syncthreads();
      const int ncand = ctl->ncand;
      double best = 0.0;
      int bestp = 0;
      auto exact = [&](int q) {  // row-order sums, bit-identical to the reference's sum(x[s::q])
        const double v = wave_max(wave_partial<T, true, LW>(work, N, q, geom[q], lane));
        if (v > best || (v == best && bestp != 0 && q < bestp)) {
          best = v;
          bestp = q;
        }
      };
      if (ncand <= kCandCap) {
        if (wv == 0) {  // survivors: upper bound reaches the final lower bound (compacted in place)
          const double lfin = __longlong_as_double((long long)ctl->lbits);
          int ns = 0;
          for (int b = 0; b < ncand; b += kWave) {
            const int k = b + lane;
            const bool keep = k < ncand && cand_hi[k] >= lfin;
            const int q = k < ncand ? cand_q[k] : 0;
            const unsigned long long m = __ballot(keep);
            const int pos = ns + __popcll(m & ((1ull << lane) - 1ull));
            if (keep) cand_q[pos] = q;
            ns += __popcll(m);
          }
          if (lane == 0) ctl->nsurv = ns;
        }
        __syncthreads();
        const int ns = ctl->nsurv;
        for (int k = wv; k < ns; k += nw) exact(cand_q[k]);
      } else {  // list overflow (many near-ties): every period exactly
        for (int q = 2 + wv; q <= max_length - 1; q += nw) exact(q);
      }
      if (!(best > 0.0)) bestp = 0;
      if (lane == 0) {
        wbest[wv] = best;
        wbestp[wv] = bestp;
      }
      __syncthreads();
      red_argmax(wbest, wbestp, nw, best, bestp);
      __syncthreads();
      if (bestp == 0) {
        status = 1;  // reference: project(data, None) raises
      } else {
        // project, subtract unconditionally (:334-340)
        if (!general) {
          const Fold f(N, bestp);
          for (int j = tid; j < bestp; j += blockDim.x) {
            const T m = residue_mean(work, f, j, false);
            const int cnt = f.count(j);
            for (int r = 0; r < cnt; ++r) {
              const int n = r * bestp + j;
              brow[n] = m;
              work[n] -= m;
            }
          }
        } else {
          project_lds(work, buf, N, bestp, flags, tb);
          for (int n = tid; n < N; n += blockDim.x) {
            const T m = buf[n];
            brow[n] = m;
            work[n] -= m;
          }
        }
        __syncthreads();
        const double this_norm = sqrt(block_sumsq(work, N, red)) / sqrtN;
        const double gain = (old_norm - this_norm) / og;
        if (gain > ratio) {  // :343
          keep_p = (uint32_t)bestp;
          keep_n = gain;
          old_norm = this_norm;
          zero_row = false;
        }
      }
    }
    if (zero_row)
      for (int n = tid; n < N; n += blockDim.x) brow[n] = T(0);
    if (tid == 0) {
      periods_out[w * num + i] = keep_p;
      norms_out[w * num + i] = keep_n;
    }
  }
  if (tid == 0) status_out[w] = status;
}

// ======================================================================================
// Periods.best_correlation, window-pair screen (plain projection, fp64 windows that fit the LDS twice).
//   Same layout as k_mbest_step1_pair: two windows per workgroup, pair window + one fp64 staging buffer in LDS, fp64
//   residuals in the HBM workspace.  Per outer iteration (Periods.py:320-347) the multi-period passes run over the
//   float images with "largest square" in place of "sum of squares" (max_s |S_p[s]|, :327-331); |sqrt(screen) - max|S||
//   <= 1.25 (R + 2) 2^-24 sum|r| (any summation order of R float-rounded terms), so the periods whose upper bound
//   reaches the best lower bound contain the exact argmax; they are re-evaluated with the row-order fp64 pass of
//   k_best_correlation and compared exactly as there (strict '>', lowest period among equals).  Both windows of a pair
//   always run the same number of iterations, so they never fall out of step.
// ======================================================================================
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PH_STEP1_WAVES, 8))) void k_best_correlation_pair(
    const double* __restrict__ x, int W, int N, int num, int max_length, double ratio, const PGeom* __restrict__ geom,
    const PGeomF* __restrict__ geomf, const PassPlan* __restrict__ plan, int n_pass, double* __restrict__ gres,
    uint32_t* __restrict__ periods_out, double* __restrict__ norms_out, double* __restrict__ bases_out,
    int* __restrict__ status_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  f2* pw = cv.take<f2>(N + kPad);
  double* stg = cv.take<double>(N + kPad);
  double* red = cv.take<double>(kRedDoubles);
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  int* list = cv.take<int>(2 * kPairListCap);
  int* ctl = cv.take<int>(8);        // [w] survivors listed, [2+w] status, [4+w] window exists, [6] pass queue
  double* st = cv.take<double>(10);  // [w] sum|r| (scaled), [2+w] scale, [4+w] og, [6+w] old_norm, [8+w] usable (1 / 0)

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const size_t gstride = win_stride((size_t)N);
  float* pwf = reinterpret_cast<float*>(pw);
  const double sqrtN = uniform_f64(sqrt((double)N));
  const int p_lo = 2, p_hi = max_length - 1, P = p_hi - p_lo + 1;

  zero_pad(stg, N);
  for (int i = tid; i < kPad; i += blockDim.x) pw[N + i] = f2_zero();
  for (int w = 0; w < 2; ++w) {
    const int64_t gw = 2 * (int64_t)blockIdx.x + w;
    const bool exists = gw < W;
    __syncthreads();
    if (exists) {
      load_window(x + gw * (int64_t)N, stg, N);
    } else {
      for (int n = tid; n < N; n += blockDim.x) stg[n] = 0.0;
    }
    __syncthreads();
    double a2 = 0.0, a1 = 0.0;
    for (int n = tid; n < N; n += blockDim.x) {
      const double v = stg[n];
      a2 = fma(v, v, a2);
      a1 += fabs(v);
    }
    block_sum2(a2, a1, red);
    const double sc = uniform_f64(pair_pick_scale(a2, N));
    for (int n = tid; n < N; n += blockDim.x) pwf[2 * n + w] = (float)(stg[n] * sc);
    if (tid == 0) {
      st[w] = a1 * sc * (1.0 + 1e-9);
      st[2 + w] = sc;
      st[4 + w] = st[6 + w] = sqrt(a2) / sqrtN;  // og, old_norm (Periods.py:316-317)
      st[8 + w] = pair_usable(a2) ? 1.0 : 0.0;
      ctl[2 + w] = 0;
      ctl[4 + w] = exists ? 1 : 0;
      ctl[6] = nw;  // pass queue of the screen
    }
  }
  __syncthreads();

  for (int it = 0; it < num; ++it) {
    const bool run0 = ctl[4] && ctl[2] == 0, run1 = ctl[5] && ctl[3] == 0;
    __syncthreads();
    if (tid == 0) ctl[0] = ctl[1] = 0;
    prio_by_progress(it);
    if ((run0 || run1) && P > 0) {
      // ---- screen: largest square of a residue sum per period, both windows (Periods.py:324-331 in float)
      f2* vals = reinterpret_cast<f2*>(stg);
      pair_sweep_plan<true>(
          pw, N, geomf, plan, wv, n_pass, nw, [&](float z, int q) { pair_store1(vals, z, q, p_lo); },
          [&](float z, int q_a, int q_b) { pair_store2(vals, z, q_a, q_b, p_lo); },
          PH_PAIR_QUEUE ? &ctl[6] : nullptr);
      prio_short_phase();
      __syncthreads();
      if (tid == 0) ctl[6] = nw;  // for the next screen (barriers in between)
      const bool scr0 = run0 && st[8] != 0.0, scr1 = run1 && st[9] != 0.0;
      if (scr0 || scr1) {
        const double u0 = 1.25 * 5.9604644775390625e-08 * st[0], u1 = 1.25 * 5.9604644775390625e-08 * st[1];
        const float fn = (float)N;
        double lo0 = -1.0 / 0.0, lo1 = -1.0 / 0.0;
        for (int idx = tid; idx < P; idx += blockDim.x) {
          const f2 v = vals[idx];
          const double rr = (double)(pair_rows_upper(fn, p_lo + idx) + 2);
          lo0 = fmax(lo0, sqrt((double)v.x) - rr * u0);
          lo1 = fmax(lo1, sqrt((double)v.y) - rr * u1);
        }
        lo0 = wave_max(lo0);
        lo1 = wave_max(lo1);
        if (lane == 0) {
          red[wv] = lo0;
          red[kMaxWaves + wv] = lo1;
        }
        __syncthreads();
        lo0 = red[0];
        lo1 = red[kMaxWaves];
        for (int i = 1; i < nw; ++i) {
          lo0 = fmax(lo0, red[i]);
          lo1 = fmax(lo1, red[kMaxWaves + i]);
        }
        for (int idx = tid; idx < P; idx += blockDim.x) {
          const f2 v = vals[idx];
          const double rr = (double)(pair_rows_upper(fn, p_lo + idx) + 2);
          const double h0 = sqrt((double)v.x) + rr * u0, h1 = sqrt((double)v.y) + rr * u1;
          if (scr0 && h0 > 0.0 && h0 >= lo0) {
            const int k = atomicAdd(&ctl[0], 1);
            if (k < kPairListCap) list[k] = p_lo + idx;
          }
          if (scr1 && h1 > 0.0 && h1 >= lo1) {
            const int k = atomicAdd(&ctl[1], 1);
            if (k < kPairListCap) list[kPairListCap + k] = p_lo + idx;
          }
        }
      }
    }
    __syncthreads();
    for (int w = 0; w < 2; ++w) {
      const int64_t gw = 2 * (int64_t)blockIdx.x + w;
      if (!ctl[4 + w]) continue;
      double* brow = bases_out + (gw * num + it) * (int64_t)N;
      int status = ctl[2 + w];
      uint32_t keep_p = 0u;
      double keep_n = 0.0;
      bool zero_row = true;
      const double og = st[4 + w], old_norm = st[6 + w], sc = st[2 + w];
      const bool exact_all = st[8 + w] == 0.0 || ctl[w] > kPairListCap;
      const int ncand = exact_all ? P : ctl[w];
      __syncthreads();
      if (status == 0) {
        const double* src = it == 0 ? x + gw * (int64_t)N : gres + gw * gstride;
        load_window(src, stg, N);
        __syncthreads();
        double best = 0.0;
        int bestp = 0;
        for (int k = wv; k < ncand; k += nw) {  // row-order sums, bit-identical to the reference's sum(x[s::q])
          const int q = exact_all ? p_lo + k : list[w * kPairListCap + k];
          const double v = wave_max(wave_partial<double, true, true>(stg, N, q, geom[q], lane));
          if (v > best || (v == best && bestp != 0 && q < bestp)) {
            best = v;
            bestp = q;
          }
        }
        if (!(best > 0.0)) bestp = 0;
        if (lane == 0) {
          wbest[wv] = best;
          wbestp[wv] = bestp;
        }
        __syncthreads();
        red_argmax(wbest, wbestp, nw, best, bestp);
        __syncthreads();
        if (bestp == 0) {
          status = 1;  // reference: project(data, None) raises
        } else {
          // project, subtract unconditionally (:334-340); the new residual goes to the workspace and, as floats, into pw
          double* dst = gres + gw * gstride;
          double a2 = 0.0, a1 = 0.0;
          const Fold f(N, bestp);
          for (int j = tid; j < bestp; j += blockDim.x) {
            const double m = residue_mean(stg, f, j, false);
            const int cnt = f.count(j);
            for (int r = 0; r < cnt; ++r) {
              const int n = r * bestp + j;
              const double v = stg[n] - m;
              brow[n] = m;
              dst[n] = v;
              pwf[2 * n + w] = (float)(v * sc);
              a2 = fma(v, v, a2);
              a1 += fabs(v);
            }
          }
          block_sum2(a2, a1, red);
          const double this_norm = sqrt(a2) / sqrtN;
          const double gain = (old_norm - this_norm) / og;
          zero_row = !(gain > ratio);  // :343
          if (!zero_row) {
            keep_p = (uint32_t)bestp;
            keep_n = gain;
          }
          double sc2 = sc;
          const bool ok = pair_usable(a2);
          if (ok && a2 * sc * sc < 9.0e-13 * (double)N) {  // float image below 2^-20 RMS: renew its scale
            sc2 = uniform_f64(pair_pick_scale(a2, N));
            const float up = (float)(sc2 / sc);
            __syncthreads();
            for (int n = tid; n < N; n += blockDim.x) pwf[2 * n + w] *= up;
          }
          if (tid == 0) {
            st[w] = a1 * sc2 * (1.0 + 1e-9);
            st[2 + w] = sc2;
            if (!zero_row) st[6 + w] = this_norm;
            st[8 + w] = ok ? 1.0 : 0.0;
          }
        }
      }
      if (zero_row) {
        __syncthreads();  // (brow may hold this round's projection)
        for (int n = tid; n < N; n += blockDim.x) brow[n] = 0.0;
      }
      if (tid == 0) {
        periods_out[gw * num + it] = keep_p;
        norms_out[gw * num + it] = keep_n;
        ctl[2 + w] = status;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (tid < 2) {
    const int64_t gw = 2 * (int64_t)blockIdx.x + tid;
    if (gw < W) status_out[gw] = ctl[2 + tid];
  }
}

// ======================================================================================
// Periods.best_frequency  (Periods.py:351-398).  Two launches per round:
//   k_bf_spectrum: grid (bin chunks, W).  The residual is staged in LDS; thread per rfft bin k,
//     X[k] = sum_n x[n] (cos - i sin)(2 pi k n / L) with the phase index k n mod L kept
//     incrementally and the twiddles read from a float64 table (L2-resident); every workgroup
//     leaves its best (|X|^2, k), first maximum.  A single window still fills the chip.
//   k_bf_update: one workgroup per window: argmax over the chunks, p = rint(2 L / k), project
//     (all flag combinations), store, subtract, residual back to the HBM workspace.
// ======================================================================================
constexpr int kBfBlock = 256;

template <typename T, bool LW>
__global__ __launch_bounds__(kBfBlock) void k_bf_spectrum(const T* __restrict__ res, int N, int L,
                                                          const double2* __restrict__ tw,
                                                          const int* __restrict__ status,
                                                          double* __restrict__ part_m2, int* __restrict__ part_k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  const int64_t w = blockIdx.y;
  const T* xs = res + w * (int64_t)N;  // LW == false: every thread walks the residual in HBM / L2 (broadcast reads)
  T* stage = LW ? cv.take<T>(N) : nullptr;
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  const int chunk = blockIdx.x, nchunk = gridDim.x;
  if (status[w] != 0) return;  // the reference has raised for this window already
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int M = N < L ? N : L;  // rfft(data, L) truncates or zero-pads to L samples
  if constexpr (LW) {
    load_window(xs, stage, N);
    __syncthreads();
    xs = stage;
  }
  double best = -1.0;
  int bestk = 0;  // bin + 1; 0 = none
  const int k = chunk * (int)blockDim.x + tid;
  if (k <= L / 2) {
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int n = 0; n < M; ++n) {
      const double xv = (double)xs[n];
      const double2 cs = tw[idx];
      re = fma(xv, cs.x, re);
      im = fma(xv, cs.y, im);
      idx += k;
      if (idx >= L) idx -= L;
    }
    const double m2 = re * re + im * im;
    if (m2 > best) {  // false for NaN
      best = m2;
      bestk = k + 1;
    }
  }
  wave_argmax(best, bestk);
  if (lane == 0) {
    wbest[wv] = best;
    wbestp[wv] = bestk;
  }
  __syncthreads();
  if (tid == 0) {
    best = -1.0;
    bestk = 0;
    for (int i = 0; i < nw; ++i)
      if (wbestp[i] != 0 && (bestk == 0 || wbest[i] > best || (wbest[i] == best && wbestp[i] < bestk))) {
        best = wbest[i];
        bestk = wbestp[i];
      }
    part_m2[w * nchunk + chunk] = best;
    part_k[w * nchunk + chunk] = bestk;
  }
}

// In-LDS radix-2 FFT stages (decimation in time; the data is already in bit-reversed order):
// log2 L stages of L / 2 butterflies, twiddles exp(-2 pi i j / 2^s) from a float64 (cos, sin) table
// of the L-th roots.  Ends with a barrier.
__device__ __forceinline__ void lds_fft_stages(double* __restrict__ re, double* __restrict__ im, int L, int logL,
                                               const double2* __restrict__ tw) {
  for (int s = 1; s <= logL; ++s) {
    const int half = 1 << (s - 1);
    const int tstep = L >> s;
    for (int b = threadIdx.x; b < (L >> 1); b += blockDim.x) {
      const int j = b & (half - 1);
      const int a0 = ((b >> (s - 1)) << s) + j, a1 = a0 + half;
      const double2 cs = tw[j * tstep];  // (cos, sin) of +angle; the forward transform uses cos - i sin
      const double xr = re[a1], xi = im[a1];
      const double tr = fma(xr, cs.x, xi * cs.y), ti = fma(xi, cs.x, -xr * cs.y);
      const double ur = re[a0], ui = im[a0];
      re[a0] = ur + tr;
      im[a0] = ui + ti;
      re[a1] = ur - tr;
      im[a1] = ui - ti;
    }
    __syncthreads();
  }
}

// First maximum of re[k]^2 + im[k]^2 over k <= kmax -> the (|X|^2, bin + 1) record of window w, slot 0.
__device__ __forceinline__ void bf_store_peak(const double* __restrict__ re, const double* __restrict__ im, int kmax,
                                              double* wbest, int* wbestp, int64_t w, double* __restrict__ part_m2,
                                              int* __restrict__ part_k) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  double best = -1.0;
  int bestk = 0;  // bin + 1; 0 = none
  for (int k = tid; k <= kmax; k += blockDim.x) {
    const double m2 = re[k] * re[k] + im[k] * im[k];
    if (m2 > best) {  // false for NaN; ascending k per thread keeps the first maximum
      best = m2;
      bestk = k + 1;
    }
  }
  wave_argmax(best, bestk);
  if (lane == 0) {
    wbest[wv] = best;
    wbestp[wv] = bestk;
  }
  __syncthreads();
  if (tid == 0) {
    best = -1.0;
    bestk = 0;
    for (int i = 0; i < nw; ++i)
      if (wbestp[i] != 0 && (bestk == 0 || wbest[i] > best || (wbest[i] == best && wbestp[i] < bestk))) {
        best = wbest[i];
        bestk = wbestp[i];
      }
    part_m2[w] = best;
    part_k[w] = bestk;
  }
}

// Power-of-two win_size that fits the LDS: the spectrum is an in-LDS radix-2 FFT, one workgroup per
// window -- O(L log L) instead of the O(N L) of the direct DFT above.  Leaves the same
// (|X|^2, bin + 1) record, first maximum, in chunk slot 0.
template <typename T>
__global__ __launch_bounds__(kBlockWide) void k_bf_fft(const T* __restrict__ res, int N, int L, int logL,
                                                       const double2* __restrict__ tw,
                                                       const int* __restrict__ status,
                                                       double* __restrict__ part_m2, int* __restrict__ part_k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  double* re = cv.take<double>(L);
  double* im = cv.take<double>(L);
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  const int64_t w = blockIdx.x;
  if (status[w] != 0) return;  // the reference has raised for this window already
  const int M0 = N < L ? N : L;  // rfft(data, L) truncates or zero-pads to L samples
  const T* xs = res + w * (int64_t)N;
  for (int n = threadIdx.x; n < L; n += blockDim.x) {
    const int r = (int)(__brev((unsigned)n) >> (32 - logL));
    re[r] = n < M0 ? (double)xs[n] : 0.0;
    im[r] = 0.0;
  }
  __syncthreads();
  lds_fft_stages(re, im, L, logL, tw);
  bf_store_peak(re, im, L >> 1, wbest, wbestp, w, part_m2, part_k);
}

// Any other win_size whose chirp convolution fits the LDS: Bluestein.  With w[n] = exp(-i pi n^2 / L),
//   X[k] = w[k] sum_n (x[n] w[n]) conj(w[k - n]),
// a convolution evaluated with two radix-2 FFTs of size M >= min(N, L) + L/2 + 1 (only the bins k <= L/2
// are needed, so M is about 1.5 L instead of 2 L): a = x w (zero-padded) -> FFT -> times B = FFT(conj(w),
// wrapped; built once per (L, N) on the host) -> inverse FFT.  |X[k]| = |c[k]|, the final chirp is skipped.
template <typename T>
__global__ __launch_bounds__(kBlockWide) void k_bf_chirp(const T* __restrict__ res, int N, int L, int M, int logM,
                                                         const double2* __restrict__ twm,
                                                         const double2* __restrict__ chirp,
                                                         const double2* __restrict__ bfft,
                                                         const int* __restrict__ status,
                                                         double* __restrict__ part_m2, int* __restrict__ part_k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  double* re = cv.take<double>(M);
  double* im = cv.take<double>(M);
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  const int64_t w = blockIdx.x;
  if (status[w] != 0) return;
  const int M0 = N < L ? N : L;
  const T* xs = res + w * (int64_t)N;
  for (int n = threadIdx.x; n < M; n += blockDim.x) {
    const int r = (int)(__brev((unsigned)n) >> (32 - logM));
    double ar = 0.0, ai = 0.0;
    if (n < M0) {
      const double xv = (double)xs[n];
      const double2 cs = chirp[n];  // (cos, sin)(pi n^2 / L); w[n] = cos - i sin
      ar = xv * cs.x;
      ai = -xv * cs.y;
    }
    re[r] = ar;
    im[r] = ai;
  }
  __syncthreads();
  lds_fft_stages(re, im, M, logM, twm);
  // pointwise product with B, conjugated for the inverse transform (ifft(z) = conj(fft(conj(z))) / M), and
  // moved to bit-reversed order: element n and its mirror are handled by the thread that owns min(n, brev(n))
  for (int n = threadIdx.x; n < M; n += blockDim.x) {
    const int r = (int)(__brev((unsigned)n) >> (32 - logM));
    if (n > r) continue;
    const double2 b0 = bfft[n];
    const double p0r = re[n] * b0.x - im[n] * b0.y, p0i = re[n] * b0.y + im[n] * b0.x;
    if (r == n) {
      re[n] = p0r;
      im[n] = -p0i;
    } else {
      const double2 b1 = bfft[r];
      const double p1r = re[r] * b1.x - im[r] * b1.y, p1i = re[r] * b1.y + im[r] * b1.x;
      re[r] = p0r;
      im[r] = -p0i;
      re[n] = p1r;
      im[n] = -p1i;
    }
  }
  __syncthreads();
  lds_fft_stages(re, im, M, logM, twm);
  // c[k] = conj(result[k]) / M: the magnitudes only differ by the common factor 1 / M (kept out: the record is
  // only compared with the other bins of the same window)
  bf_store_peak(re, im, L >> 1, wbest, wbestp, w, part_m2, part_k);
}

template <typename T, bool LW>
__global__ __launch_bounds__(kBlockWide) void k_bf_update(T* __restrict__ res, int N, int L, int num, int it,
                                                          unsigned flags, Tables tb, T* __restrict__ gbuf, int nchunk,
                                                          const double* __restrict__ part_m2,
                                                          const int* __restrict__ part_k, double* __restrict__ dnorm,
                                                          uint32_t* __restrict__ periods_out,
                                                          double* __restrict__ powers_out, T* __restrict__ bases_out,
                                                          int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  const int64_t w = blockIdx.x;
  // LW == false: the residual is projected where it lives (the HBM workspace `res`)
  T* work = LW ? cv.take<T>(N) : res + w * (int64_t)N;
  T* buf = gbuf ? gbuf + (int64_t)blockIdx.x * N : cv.take<T>(N);
  double* red = cv.take<double>(kRedDoubles);
  const int tid = threadIdx.x;
  T* brow = bases_out + (w * num + it) * (int64_t)N;
  bool dead = status[w] != 0;
  int p = 0;
  if (!dead) {
    double best = -1.0;
    int bestk = 0;
    for (int c = 0; c < nchunk; ++c) {  // chunks in ascending bin order: strict '>' keeps the first maximum
      const double v = part_m2[w * nchunk + c];
      const int kk = part_k[w * nchunk + c];
      if (kk != 0 && (bestk == 0 || v > best)) {
        best = v;
        bestk = kk;
      }
    }
    if (bestk <= 1) {  // bin 0 (or nothing comparable): 2 L / 0 in the reference
      dead = true;
      __syncthreads();  // every thread has read status[w]
      if (tid == 0) status[w] = 1;
    } else {
      p = (int)rint(2.0 * (double)L / (double)(bestk - 1));
    }
  }
  if (dead) {  // rows from the failing round on stay zero
    for (int n = tid; n < N; n += blockDim.x) brow[n] = T(0);
    if (tid == 0) {
      periods_out[w * num + it] = 0u;
      powers_out[w * num + it] = 0.0;
    }
    return;
  }
  if constexpr (LW) {
    load_window(res + w * (int64_t)N, work, N);
    __syncthreads();
  }
  double dn;
  if (it == 0) {
    dn = periodic_norm_from_sq(block_sumsq(work, N, red), N, 0);
    if (tid == 0) dnorm[w] = dn;
  } else {
    dn = dnorm[w];
  }
  project_lds(work, buf, N, p, flags, tb);
  const double nrm = periodic_norm_from_sq(block_sumsq(buf, N, red), N, 0);
  for (int n = tid; n < N; n += blockDim.x) {
    const T b = buf[n];
    brow[n] = b;
    res[w * (int64_t)N + n] = work[n] - b;
  }
  if (tid == 0) {
    periods_out[w * num + it] = (uint32_t)p;
    powers_out[w * num + it] = nrm / dn;
  }
}

// ======================================================================================
// RamanujanPeriods.find_periods  (RamanujanPeriods.py:67-86, :124-169), folded form.
//   Rows of the reference dictionary are shifts of c_q / phi(q) (the row / max(row) at :129
//   cancels the L2 normalisation of :168).  With S = fold of x to period q:
//     a = S (star) c_q / phi,  out = a (*) c_q / phi  ==  (q/phi)^2 * E_q S,
//   where E_q = sum_{d | q} mu(q/d) P_d = prod_{prime r | q} (I - P_{q/r}) is the projector onto
//   the Ramanujan subspace (P_d = mean over the q/d cosets mod d), and
//   norms[q] = sum_j cnt_q[j] * out_j^2.
//
//   Hierarchical folds.  Only the periods of the upper half of the range, Q in (q_hi/2, q_hi]
//   ("roots"), are folded from the window (N reads each, chunked wave fold into the wave's LDS
//   strip A).  Every smaller period q has a multiple Q among the roots and S_q[i] = sum_t
//   S_Q[i + t q] is a fold of the STRIP (Q reads instead of N): the host hands every q <= q_hi/2
//   to one root ("children").  Children are folded from A into the scratch strip B and filtered
//   there; the root itself is filtered in place last.
//
//   The projector is applied one prime factor at a time, in place.  A step with d = q/r cosets
//   uses lane-per-coset when d >= 64 and a row-split mapping (64/d groups of d lanes, shuffle
//   tree) when d < 64, so that a large prime factor never serialises a coset in one lane.
//   One wavefront works on one root at a time; the wavefronts of a workgroup share the window.
// ======================================================================================
constexpr int kRamMaxWaves = 16;
#ifndef PH_RAM_U
#define PH_RAM_U 4
#endif
#ifndef PH_RAM_FOLD_V1
#define PH_RAM_FOLD_V1 0  // 1: the round-3 root fold (blocks of PH_RAM_U rows, leftover rows one at a time)
#endif

// Everything the kernel needs to know about one period, in ONE 128-byte record (two s_load_dwordx16): the strip
// steps of a wavefront are chains of dependent LDS round trips with 3-4 wavefronts per SIMD to hide them, and the
// round-2 tables (root -> child list -> step offsets -> steps -> row-split geometry, geometry, scale) put five
// dependent scalar loads from memory in front of every period.
struct RamStep {  // one factor (I - P_d) of E_q: d = q / r cosets of r elements
  int dr;         // d | r << 16
  int sm;         // row-split geometry of d < 64: G | inv16 << 8, G = 64 / d row groups, inv16 = ceil(65536 / d)
  double inv_r;   // (lane / d == (lane * inv16) >> 16 for lane < 64)
};
constexpr int kRamMaxSteps = 5;  // distinct primes of q < 30030
struct RamJob {
  int q, k, nfull, rows;  // period; k = Q / q rows of a child in its root's strip; geometry of the fold of the window
  int c0, c1, flags, sm;  // root: its children are jobs [c0, c1); flags = emit | nsteps << 8; sm: geometry of q < 64
  double scale2, pad;     // (q / phi(q))^2
  RamStep st[kRamMaxSteps];
};
static_assert(sizeof(RamJob) == 128, "RamJob is uploaded as 32 words");

// Column sums of C chunks of the fold of the window to period p, stored into the strip.  (Taking the rows that do
// not fill a block of U and the partial last row as ONE more batch of U x C loads, absent rows read from the zeroed
// pad, was slower: 2.47 -> 2.59 ms at config 3 -- the batch always issues U x C loads, and the fold is paced by the
// instructions it issues, not by the round trips of its tail.)
template <typename T, int C, int U, bool LW>
__device__ __forceinline__ void fold_store_group(const T* __restrict__ xs, int p, int rows, int nfull, int c0,
                                                 int lane, double* __restrict__ sbuf) {
  double s[C];
#pragma unroll
  for (int c = 0; c < C; ++c) s[c] = 0.0;
  fold_rows<T, C, U, LW>(Win<T, LW>::cast(xs) + lane + 64 * c0, p, rows - 1, s);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = 64 * (c0 + c) + lane;
    const bool has = j < nfull;
    const T v = xs[has ? (rows - 1) * p + j : 0];
    if (j < p) sbuf[j] = s[c] + (has ? (double)v : 0.0);
  }
}

// Root fold without remainder rows (round 4).  fold_store_group above takes the rows of a residue in blocks of U and
// then drains after every single leftover row and after the partial last row: at config 3 (22 rows of 6 chunks per
// root) a group issued 5 full batches and up to 4 one-row batches, each a full LDS round trip with 4 wavefronts per
// SIMD to hide it -- the folds ran at 0.55 of the streaming LDS rate although the bare 16-loads / drain / 16-adds
// pattern reaches 0.85 at that occupancy (tools/micro/lds_pipe_bench 1024 163840).  Here the `rows` rows of a root are
// cut into k = ceil(rows / 8) batches of U or U + 1 rows (U = rows / k, compile time), every batch fully in flight
// before its one wait, and the partial last row is the last row of the last batch (lanes without it read element 0
// and add nothing).  The order of the additions per residue is unchanged (row 0, 1, ..., rows - 1).
template <typename T, int C, int NR, bool LAST, bool LW>
__device__ __forceinline__ void ram_fold_batch(typename Win<T, LW>::ptr ptr, typename Win<T, LW>::ptr x0, int p, const bool (&has)[C],
                                               double (&s)[C]) {
  T v[NR][C];
#pragma unroll
  for (int u = 0; u < NR; ++u)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (LAST && u == NR - 1)
        v[u][c] = has[c] ? ptr[u * p + 64 * c] : x0[0];
      else
        v[u][c] = ptr[u * p + 64 * c];
    }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < NR; ++u)
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] += (LAST && u == NR - 1) ? (has[c] ? (double)v[u][c] : 0.0) : (double)v[u][c];
}

template <typename T, int C, int U, bool LW>
__device__ __forceinline__ void ram_fold_group(const T* __restrict__ xs, int p, int nfull, int k, int rem, int c0, int lane,
                                               double* __restrict__ sbuf) {
  typedef typename Win<T, LW>::ptr wptr;
  double s[C];
  bool has[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    s[c] = 0.0;
    has[c] = 64 * (c0 + c) + lane < nfull;
  }
  const wptr x0 = Win<T, LW>::cast(xs);
  wptr ptr = x0 + lane + 64 * c0;
  int b = 0;
  for (; b < rem; ++b) {  // rem < k batches of U + 1 rows
    ram_fold_batch<T, C, U + 1, false, LW>(ptr, x0, p, has, s);
    ptr += (U + 1) * p;
  }
  for (; b + 1 < k; ++b) {
    ram_fold_batch<T, C, U, false, LW>(ptr, x0, p, has, s);
    ptr += U * p;
  }
  ram_fold_batch<T, C, U, true, LW>(ptr, x0, p, has, s);  // ends with the partial row
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = 64 * (c0 + c) + lane;
    if (j < p) sbuf[j] = s[c];
  }
}

template <typename T, int U, bool LW>
__device__ __forceinline__ void ram_root_fold_u(const T* __restrict__ xs, int Q, int nfull, int k, int rem, int lane,
                                                double* __restrict__ sbuf) {
  const int nchunks = (Q + 63) >> 6;
  int k0 = 0;
  for (; k0 + 4 <= nchunks; k0 += 4) ram_fold_group<T, 4, U, LW>(xs, Q, nfull, k, rem, k0, lane, sbuf);
  switch (nchunks - k0) {
    case 3: ram_fold_group<T, 3, U, LW>(xs, Q, nfull, k, rem, k0, lane, sbuf); break;
    case 2: ram_fold_group<T, 2, U, LW>(xs, Q, nfull, k, rem, k0, lane, sbuf); break;
    case 1: ram_fold_group<T, 1, U, LW>(xs, Q, nfull, k, rem, k0, lane, sbuf); break;
    default: break;
  }
}

// S_Q of the window into the strip, Q >= 64 (rows = ceil(N / Q) >= 1)
template <typename T, bool LW>
__device__ __forceinline__ void ram_root_fold(const T* __restrict__ xs, int Q, int rows, int nfull, int lane,
                                              double* __restrict__ sbuf) {
  const int k = (rows + 7) >> 3;  // batches
  int U = 1;                      // rows / k without a division (k <= rows, the quotient is in 1 .. 8)
  while ((U + 1) * k <= rows) ++U;
  const int rem = rows - U * k;
  switch (U) {
    case 1: ram_root_fold_u<T, 1, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
    case 2: ram_root_fold_u<T, 2, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
    case 3: ram_root_fold_u<T, 3, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
    case 4: ram_root_fold_u<T, 4, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
    case 5: ram_root_fold_u<T, 5, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
    case 6: ram_root_fold_u<T, 6, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
    case 7: ram_root_fold_u<T, 7, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
    default: ram_root_fold_u<T, 8, LW>(xs, Q, nfull, k, rem, lane, sbuf); break;
  }
}

// Row-split coset sums of s[0 .. len) modulo d < 64 (d | len): lane = g d + i (g < G = 64 / d) adds
// s[lane + k G d]; the G partial sums of a coset are combined with a shuffle tree.  Returns the coset
// total in EVERY lane of the coset's column (lane mod d), 0 in the idle lanes >= G d.
__device__ __forceinline__ double strip_cosets_small(const double* __restrict__ s, int len, int d, int sm, int lane) {
  const int G = sm & 255, inv16 = sm >> 8, L = G * d;
  double part = 0.0;
  if (lane < L) {
    int idx = lane;
    for (; idx + 3 * L < len; idx += 4 * L) {  // four loads in flight; the sum keeps its order
      const double a = s[idx], b = s[idx + L], c = s[idx + 2 * L], e = s[idx + 3 * L];
      part += a;
      part += b;
      part += c;
      part += e;
    }
    for (; idx < len; idx += L) part += s[idx];
  }
  int sft = 1;
  while (sft < G) sft <<= 1;
  for (sft >>= 1; sft >= 1; sft >>= 1) {
    const int src = lane + sft * d;
    const double o = __shfl(part, src & (kWave - 1), kWave);
    part += (src < L) ? o : 0.0;
  }
  const int g = (lane * inv16) >> 16;                     // lane / d
  const double tot = __shfl(part, lane - g * d, kWave);  // the column's total sits in its first row group
  return lane < L ? tot : 0.0;
}

// The strip loops below run with compile-time trip counts where the counts are small (a coset of r = 2, 3, 5, 7
// elements when d >= 64 and q <= 512; a child with k = Q / q <= 8 rows) and take two lane chunks per trip: the
// kernel is paced by the dependent LDS round trips of each wavefront, so all loads of a trip are in flight
// together.  Lanes past the end read element 0 and write nothing.  Sums keep the order of the generic loops.
template <int K>
__device__ __forceinline__ void strip_fold_fixed(const double* __restrict__ src, int q, double* __restrict__ dst,
                                                 int lane) {
  for (int c0 = 0; c0 < q; c0 += 2 * kWave) {
    const int i0 = c0 + lane, i1 = i0 + kWave;
    const bool ok0 = i0 < q, ok1 = i1 < q;
    const double* p0 = src + (ok0 ? i0 : 0);
    const double* p1 = src + (ok1 ? i1 : 0);
    double v0[K], v1[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      v0[t] = p0[t * q];
      v1[t] = p1[t * q];
    }
    double e0 = 0.0, o0 = 0.0, e1 = 0.0, o1 = 0.0;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      if (t & 1) {
        o0 += v0[t];
        o1 += v1[t];
      } else {
        e0 += v0[t];
        e1 += v1[t];
      }
    }
    if (ok0) dst[i0] = e0 + o0;
    if (ok1) dst[i1] = e1 + o1;
  }
}

// dst[i] = sum_t src[i + t q], i < q  (len = k q): the fold of a folded strip.
__device__ __forceinline__ void strip_fold(const double* __restrict__ src, int len, int q, int k, int sm,
                                           double* __restrict__ dst, int lane) {
  if (q < 64) {
    const double tot = strip_cosets_small(src, len, q, sm, lane);
    if (lane < q) dst[lane] = tot;
    return;
  }
  switch (k) {
    case 2: strip_fold_fixed<2>(src, q, dst, lane); return;
    case 3: strip_fold_fixed<3>(src, q, dst, lane); return;
    case 4: strip_fold_fixed<4>(src, q, dst, lane); return;
    case 5: strip_fold_fixed<5>(src, q, dst, lane); return;
    case 6: strip_fold_fixed<6>(src, q, dst, lane); return;
    case 7: strip_fold_fixed<7>(src, q, dst, lane); return;
    case 8: strip_fold_fixed<8>(src, q, dst, lane); return;
    default: break;
  }
  for (int i = lane; i < q; i += kWave) {
    double a0 = 0.0, a1 = 0.0;
    int t = 0;
    for (; t + 2 <= k; t += 2) {
      a0 += src[i + t * q];
      a1 += src[i + (t + 1) * q];
    }
    if (t < k) a0 += src[i + t * q];
    dst[i] = a0 + a1;
  }
}

// Sum of squares of the filtered strip, split by the count of the residue: norms[q] = (q / phi)^4 (cs all + full) with
// cs = rows - 1 (every residue has at least rows - 1 samples, the residues below nfull one more).
struct RamAcc {
  double all = 0.0, full = 0.0;
  __device__ __forceinline__ void add(double o, int j, int nfull) {
    const double t = o * o;
    all += t;
    full += j < nfull ? t : 0.0;
  }
};

// One factor (I - P_d) of the projector: every element of s[0 .. q) loses the mean of its coset modulo d (r = q / d
// elements per coset); each lane reads and writes only its own elements.  LAST: the last factor of a period writes
// nothing back -- the filtered values go straight into the sum of squares (one pass over the strip, its stores and
// one wavefront sync less per period).
template <int R, bool LAST>
__device__ __forceinline__ void strip_step_fixed(double* __restrict__ s, int d, double inv_r, int lane, int nfull,
                                                 RamAcc& acc) {
  for (int c0 = 0; c0 < d; c0 += 2 * kWave) {
    const int i0 = c0 + lane, i1 = i0 + kWave;
    const bool ok0 = i0 < d, ok1 = i1 < d;
    double* p0 = s + (ok0 ? i0 : 0);
    double* p1 = s + (ok1 ? i1 : 0);
    double v0[R], v1[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      v0[t] = p0[t * d];
      v1[t] = p1[t * d];
    }
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      m0 += v0[t];
      m1 += v1[t];
    }
    m0 *= inv_r;
    m1 *= inv_r;
    if (ok0) {
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (LAST)
          acc.add(v0[t] - m0, i0 + t * d, nfull);
        else
          p0[t * d] = v0[t] - m0;
      }
    }
    if (ok1) {
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (LAST)
          acc.add(v1[t] - m1, i1 + t * d, nfull);
        else
          p1[t * d] = v1[t] - m1;
      }
    }
  }
}

template <bool LAST>
__device__ __forceinline__ void strip_step(double* __restrict__ s, int q, const RamStep& st, int lane, int nfull,
                                           RamAcc& acc) {
  const int d = st.dr & 0xffff, r = st.dr >> 16;
  const double inv_r = st.inv_r;
  if (d < 64) {
    const int L = (st.sm & 255) * d;
    const double m = strip_cosets_small(s, q, d, st.sm, lane) * inv_r;
    if (lane < L) {
      int idx = lane;
      for (; idx + 3 * L < q; idx += 4 * L) {
        const double a = s[idx], b = s[idx + L], c = s[idx + 2 * L], e = s[idx + 3 * L];
        if (LAST) {
          acc.add(a - m, idx, nfull);
          acc.add(b - m, idx + L, nfull);
          acc.add(c - m, idx + 2 * L, nfull);
          acc.add(e - m, idx + 3 * L, nfull);
        } else {
          s[idx] = a - m;
          s[idx + L] = b - m;
          s[idx + 2 * L] = c - m;
          s[idx + 3 * L] = e - m;
        }
      }
      for (; idx < q; idx += L) {
        if (LAST)
          acc.add(s[idx] - m, idx, nfull);
        else
          s[idx] -= m;
      }
    }
    return;
  }
  switch (r) {  // r is prime; d >= 64 and q <= 512 leave 2, 3, 5, 7
    case 2: strip_step_fixed<2, LAST>(s, d, inv_r, lane, nfull, acc); return;
    case 3: strip_step_fixed<3, LAST>(s, d, inv_r, lane, nfull, acc); return;
    case 5: strip_step_fixed<5, LAST>(s, d, inv_r, lane, nfull, acc); return;
    case 7: strip_step_fixed<7, LAST>(s, d, inv_r, lane, nfull, acc); return;
    default: break;
  }
  for (int i = lane; i < d; i += kWave) {
    double m = 0.0;
    for (int t = 0; t < r; ++t) m += s[i + t * d];
    m *= inv_r;
    for (int t = 0; t < r; ++t) {
      if (LAST)
        acc.add(s[i + t * d] - m, i + t * d, nfull);
      else
        s[i + t * d] -= m;
    }
  }
}

// s holds S_q: apply E_q (in place, but for the last factor) and reduce to norms[q] (all lanes get it).  `steps` is
// the record's step list in memory: the next step is fetched (scalar load) while the current one runs.
__device__ __forceinline__ double ram_emit(double* __restrict__ s, const RamJob& J, const RamStep* __restrict__ steps,
                                           int lane) {
  const int q = J.q, nsteps = J.flags >> 8, nfull = J.nfull;
  RamAcc acc;
  RamStep cur = steps[0];
#pragma unroll 1
  for (int k = 0; k + 1 < nsteps; ++k) {
    const RamStep nxt = steps[k + 1];
    strip_step<false>(s, q, cur, lane, nfull, acc);
    ram_wave_sync();
    cur = nxt;
  }
  if (nsteps > 0) {
    strip_step<true>(s, q, cur, lane, nfull, acc);
  } else {  // q == 1: E_1 is the identity
    for (int j = lane; j < q; j += kWave) acc.add(s[j], j, nfull);
  }
  const double cs = (double)(J.rows - 1);
  return wave_sum(cs * acc.all + acc.full) * (J.scale2 * J.scale2);  // out = (q / phi)^2 E_q S, norms = sum cnt out^2
}

template <typename T, bool LW>
__global__ __launch_bounds__(kRamMaxWaves * 64) void k_ramanujan(const T* __restrict__ x, int N, int q_hi,
                                                                const RamJob* __restrict__ roots, int n_root,
                                                                const RamJob* __restrict__ children, T* gwin, int pad,
                                                                double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  // pad == 0 (the host drops the zeroed pad when that makes room for one more wavefront; roots >= 64 only): the reads
  // of a fold past the end of the window land in the strips -- they belong to lanes whose sums are discarded
  T* xs = window_buf<T, LW>(cv, gwin, N + pad);
  const int nw = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  const int lenA = q_hi, lenB = q_hi / 2 > 0 ? q_hi / 2 : 1;
  double* sA = cv.take<double>((size_t)nw * lenA) + (size_t)wv * lenA;  // this wave's root fold S_Q
  double* sB = cv.take<double>((size_t)nw * lenB) + (size_t)wv * lenB;  // scratch for one child

  const int64_t w = blockIdx.x;
#ifdef PH_CLOCKS
  const long long ck0 = clock64(), wk0 = wall_clock64();
#endif
  load_window(x + w * (int64_t)N, xs, N);
  if (pad) zero_pad(xs, N);
  double* orow = out + w * (int64_t)(q_hi + 1);
  // the root queue: orow[0] (0 is never a period) is zero on entry and is put back to zero at the end -- an LDS
  // counter would cost the 16 bytes that decide between 15 and 16 wavefronts at config 3
  int* next_root = reinterpret_cast<int*>(orow);
  __syncthreads();
#ifdef PH_CLOCKS
  const long long ck1 = clock64(), wk1 = wall_clock64();
#endif

  // A workgroup holds its CU alone (window + strips fill the LDS) and ends with its slowest wavefront: the roots are
  // taken from a queue, not dealt round-robin -- no wavefront idles while another still has whole roots to do.
  for (int i = wv; i < n_root;) {
    int nxt = 0;
    if (lane == 0) nxt = atomicAdd(next_root, 1);  // asked for now, needed after this root: the HBM round trip hides
    const RamJob R = roots[i];
    const int Q = R.q;
    // ---- S_Q from the window
    if (Q < 64) {
      const double tot = wave_fold_small(xs, N, Q, lane);  // row-split path, S_Q[lane] in the lanes below Q
      if (lane < Q) sA[lane] = tot;
    } else {
#if PH_RAM_FOLD_V1
      const int rows = R.rows, nfull = R.nfull;
      const int nchunks = (Q + 63) >> 6;
      int k0 = 0;
      for (; k0 + 4 <= nchunks; k0 += 4) fold_store_group<T, 4, PH_RAM_U, LW>(xs, Q, rows, nfull, k0, lane, sA);
      switch (nchunks - k0) {
        case 3: fold_store_group<T, 3, PH_RAM_U, LW>(xs, Q, rows, nfull, k0, lane, sA); break;
        case 2: fold_store_group<T, 2, 2 * PH_RAM_U, LW>(xs, Q, rows, nfull, k0, lane, sA); break;
        case 1: fold_store_group<T, 1, 4 * PH_RAM_U, LW>(xs, Q, rows, nfull, k0, lane, sA); break;
        default: break;
      }
#else
      // LDS capacity (window + strips) caps this kernel at 3-4 wavefronts per SIMD, so registers are plentiful:
      // batches of up to 9 rows x 4 chunks of loads in flight per wait, no leftover rows (ram_root_fold)
      ram_root_fold<T, LW>(xs, Q, R.rows, R.nfull, lane, sA);
#endif
    }
    ram_wave_sync();
#ifdef PH_RAM_FOLDS_ONLY  // profiling aid (results incomplete): the root folds alone, one strip element kept alive
    if (lane == 0) orow[Q] = sA[Q >> 1];
    ram_wave_sync();
    i = nw + __builtin_amdgcn_readfirstlane(nxt);
    continue;
#endif
    // ---- children: fold of the strip, filtered in the scratch strip
    for (int c = R.c0; c < R.c1; ++c) {
      const RamJob C = children[c];
      strip_fold(sA, Q, C.q, C.k, C.sm, sB, lane);
      ram_wave_sync();
      const double v = ram_emit(sB, C, children[c].st, lane);
      if (lane == 0) orow[C.q] = v;
      ram_wave_sync();  // sB is rewritten by the next child
    }
    if (R.flags & 1) {
      const double v = ram_emit(sA, R, roots[i].st, lane);
      if (lane == 0) orow[Q] = v;
    }
    ram_wave_sync();  // sA is rewritten by the next root
    i = nw + __builtin_amdgcn_readfirstlane(nxt);
  }
  __syncthreads();
  if (threadIdx.x == 0) orow[0] = 0.0;
#ifdef PH_CLOCKS
  if ((blockIdx.x == 7 || blockIdx.x == 2000) && threadIdx.x == 0) {
    const long long ck2 = clock64(), wk2 = wall_clock64();
    printf("k_ramanujan window %d: load %lld cycles / %lld ticks(100MHz), work %lld cycles / %lld ticks -> %.0f MHz\n", (int)w,
           ck1 - ck0, wk1 - wk0, ck2 - ck1, wk2 - wk1, 100.0 * (double)(ck2 - ck1) / (double)(wk2 - wk1));
  }
#endif
}

// ======================================================================================
// QOPeriods building blocks (QOPeriods.py:779-795): W = A x and A^T w for natural-basis A.
// ======================================================================================
template <typename T, bool LW>
__global__ __launch_bounds__(kBlock) void k_fold_sums(const T* __restrict__ x, int N,
                                                      const int* __restrict__ p_list,
                                                      const int* __restrict__ keep,
                                                      const int* __restrict__ row_off, int n_p, int stride,
                                                      double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int64_t w = blockIdx.x;
  const T* xs = x + w * (int64_t)N;  // LW == false: a window longer than the LDS is folded straight from HBM / L2
  if constexpr (LW) {
    Carve cv(smem);
    T* stage = cv.take<T>(N);
    load_window(xs, stage, N);
    __syncthreads();
    xs = stage;
  }
  double* orow = out + w * (int64_t)stride;
  for (int k = 0; k < n_p; ++k) {
    const int p = p_list[k];
    const Fold f(N, p);
    for (int j = threadIdx.x; j < keep[k]; j += blockDim.x)
      orow[row_off[k] + j] = (double)column_sum(xs, j, p, f.count(j));
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_tile_sum(const double* __restrict__ wts, int N,
                                                     const int* __restrict__ p_list,
                                                     const int* __restrict__ keep,
                                                     const int* __restrict__ row_off, int n_p, int stride,
                                                     T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* ws = reinterpret_cast<double*>(smem);
  const int64_t w = blockIdx.x;
  for (int i = threadIdx.x; i < stride; i += blockDim.x) ws[i] = wts[w * (int64_t)stride + i];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k < n_p; ++k) {  // row order of A, like np.matmul(A.T, w) sums over rows
      const int j = n % p_list[k];
      if (j < keep[k]) acc += ws[row_off[k] + j];
    }
    out[w * (int64_t)N + n] = (T)acc;
  }
}

// ======================================================================================
// QOPeriods.find_periods, non-orthogonal / update_weights=True branch (QOPeriods.py:373-596
// with :468-478, :510-522, :560-594; get_subspaces :830-840; solve_quadratic :779-796), the
// whole greedy loop on the device.  One workgroup per window.  Per iteration:
//   gamma-normalised all-p sweep of the residual (pass plan)  -> strongest period
//   rows kept for it = Euler-phi mass its divisors add to the running divisor set
//   right-hand side A x extended by the fold of the data over the new rows
//   A A^T w = A x by preconditioned conjugate gradients in LDS -- the Gram matrix (integer co-occurrence
//   counts of the indicator rows) is never formed, its rows are regenerated inside the product (qo_offdiag)
//   reconstruction A^T w, residual
// The reference stops when rms(reconstruction) <= rms(data) * thresh (default test_function)
// or when numpy.linalg.solve raises LinAlgError (here: non-positive curvature or no convergence of the
// solve, a period that adds no new rows, or a repeated period -- the cases that make the reference's
// matrix singular).  counts[w] = {periods reported, blocks in the dictionary}.
// ======================================================================================
#ifndef PH_QO_OCC
#define PH_QO_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))  // two 16-wave workgroups per CU need <= 64 VGPRs
#endif
constexpr int kQoMaxBlocks = 64;
constexpr int kQoPairTab = 16;  // dictionaries of up to this many blocks keep their pair constants in LDS

__device__ __forceinline__ int qo_gcd(int a, int b) {
  while (b != 0) {
    const int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// Off-diagonal part of one row of the Gram matrix A A^T times a vector, WITHOUT the matrix.  Row (a, i) is the
// indicator of n = i (mod p_a); its entry against row (b, j) counts the n < N with n = i (p_a) and n = j (p_b)
// (QOPeriods.py:781).  Along n = i, i + p_a, ... the residue mod p_b advances by p_a mod p_b and returns after
// cycle = p_b / gcd steps, so
//   sum_j G[(a,i),(b,j)] v_b[j] = sum_{t < min(cycle, terms)} count(t) v_b[idx_t],   count(t) = (terms-1-t) / cycle + 1,
// with terms = samples of residue i.  For N < lcm(p_a, p_b) every count is 1 and the loop is the fold of the tiled
// v_b.  The steps t = t0, t0 + ts, ... are summed (a group of `ts` lanes shares a long row).
__device__ __forceinline__ int qo_wrap(int idx, int pb) {  // idx in [0, 2 pb) -> idx mod pb
  return (int)min((unsigned)idx, (unsigned)(idx - pb));
}

// `vb` points at block b's entries inside a vector in the solver's layout: k_b entries followed by one zero, so an
// index past the kept rows is clamped onto the zero instead of being masked.
__device__ __forceinline__ double qo_offdiag(const double* __restrict__ vb, int kb, int pb, int cycle, int step, int i,
                                             int terms, int t0, int ts) {
  const int lim = cycle < terms ? cycle : terms;
  if (t0 >= lim) return 0.0;
  int idx = (int)(((unsigned)i + (unsigned)t0 * (unsigned)step) % (unsigned)pb);  // i < 2^20, t0 < 16, step < p_b < 2^20
  const int sstep = ts == 1 ? step : (int)(((unsigned)ts * (unsigned)step) % (unsigned)pb);
  if (cycle >= terms) {  // all counts are 1
    double s0 = 0.0, s1 = 0.0;
    int t = t0;
    for (; t + 3 * ts < lim; t += 4 * ts) {
      const int i1 = qo_wrap(idx + sstep, pb), i2 = qo_wrap(i1 + sstep, pb), i3 = qo_wrap(i2 + sstep, pb);
      const double v0 = vb[min(idx, kb)], v1 = vb[min(i1, kb)], v2 = vb[min(i2, kb)], v3 = vb[min(i3, kb)];
      s0 += v0;
      s1 += v1;
      s0 += v2;
      s1 += v3;
      idx = qo_wrap(i3 + sstep, pb);
    }
    for (; t < lim; t += ts) {
      s0 += vb[min(idx, kb)];
      idx = qo_wrap(idx + sstep, pb);
    }
    return s0 + s1;
  }
  const int chi = (terms - 1) / cycle + 1, tsw = (terms - 1) % cycle;
  double shi = 0.0, slo = 0.0;
  for (int t = t0; t < lim; t += ts) {
    const double v = vb[min(idx, kb)];
    if (t <= tsw)
      shi += v;
    else
      slo += v;
    idx = qo_wrap(idx + sstep, pb);
  }
  return (double)chi * shi + (double)(chi - 1) * slo;
}

template <typename T, bool LW>
__global__ __launch_bounds__(1024) PH_QO_OCC void k_qo_find(const T* __restrict__ x, int N, int num, double thresh,
                                                        int p_lo, int p_hi, const PGeom* __restrict__ geom,
                                                        const PassPlan* __restrict__ plan, int n_pass,
                                                        const int* __restrict__ phi, const int* __restrict__ div_off,
                                                        const int* __restrict__ div_q, int kcap, int overlay, T* gwin,
                                                        double* __restrict__ ws_all, uint32_t* __restrict__ periods_out,
                                                        double* __restrict__ norms_out, int* __restrict__ keeps_out,
                                                        int* __restrict__ counts_out, double* __restrict__ weights_out,
                                                        T* __restrict__ resid_out, int* __restrict__ status_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  // the residual: LDS, or (LW == false: windows that leave no room for the solver's vectors) the HBM workspace
  T* work = window_buf<T, LW>(cv, gwin, N + kPad);
  double* red = cv.take<double>(kRedDoubles);
  double* wbest = cv.take<double>(kMaxWaves);
  int* wbestp = cv.take<int>(kMaxWaves);
  int* bper = cv.take<int>(kQoMaxBlocks);       // period of dictionary block b
  int* bkeep = cv.take<int>(kQoMaxBlocks);      // rows kept for it
  int* boff = cv.take<int>(kQoMaxBlocks + 1);   // first row of block b
  double* bnorm = cv.take<double>(kQoMaxBlocks);
  uint32_t* seen = cv.take<uint32_t>((p_hi + 32) / 32);  // running divisor set R (QOPeriods.py:832-835)
  int* ptab = cv.take<int>(2 * kQoPairTab * kQoPairTab);  // (a, b): {p_b / gcd(p_a, p_b), p_a mod p_b}
  double* red3 = cv.take<double>(6 * kMaxWaves);  // wave partials of the solver's fused reduction (two parities)
  int* blg = cv.take<int>(kQoMaxBlocks);        // log2 of the lanes that share a row of block b in the product
  int* ioff = cv.take<int>(kQoMaxBlocks + 1);   // first work item of block b
  // Conjugate gradients on A A^T w = A x.  Solver layout of a vector: block b's k_b entries start at slot
  // boff[b] + b and are followed by one zero (qo_offdiag clamps indices past the kept rows onto it).
  const int kv = kcap + kQoMaxBlocks;
  double* xv = cv.take<double>(kv);  // weights; persistent: the rows of earlier steps start from their last values
  // The other vectors only live during a solve, when the residual window is dead (it is rebuilt from the data and the
  // reconstruction afterwards): they overlay the window buffer when that is large enough, which leaves room for a
  // second workgroup on the CU.
  Carve sv((LW && overlay) ? reinterpret_cast<unsigned char*>(work) : cv.base + cv.off);
  double* rv = sv.take<double>(kv);   // residual of the normal equations (starts as A x, QOPeriods.py:782)
  double* pv = sv.take<double>(kv);   // search direction
  double* qv = sv.take<double>(kv);   // A A^T pv
  double* zv = sv.take<double>(kv);   // preconditioned residual
  double* wv_ = sv.take<double>(kv);  // A A^T zv
  double* dv = sv.take<double>(kv);   // 1 / diagonal = 1 / samples of the row's residue (Jacobi preconditioner)
  int* trm = sv.take<int>(kv);        // samples of the row's residue

  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const T* data = x + w * (int64_t)N;
  double* wts = ws_all + w * 2 * (int64_t)kcap;  // last good weights (row order)
  double* rhs = wts + kcap;                      // A x of the rows found so far

  load_window(data, work, N);
  zero_pad(work, N);
  for (int k = tid; k < (p_hi + 32) / 32; k += blockDim.x) seen[k] = 0u;
  if (tid == 0) boff[0] = 0;
  __syncthreads();
  const double data_sq = block_sumsq(work, N, red);
  double recon_sq = 0.0;
  int nb = 0;        // blocks in the dictionary
  int status = 0;
  bool stopped_by_test = false;

#ifdef PH_QO_TIMERS
  long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tq0 = wall_clock64();
  int cg_iters = 0;
#define PH_QO_MARK(k)                         \
  {                                           \
    const long long now_ = wall_clock64();    \
    tq[k] += now_ - tq0;                      \
    tq0 = now_;                               \
  }
#else
#define PH_QO_MARK(k)
#endif
  for (int it = 0; it < num; ++it) {
    if (it > 0 && !(sqrt(recon_sq / N) > sqrt(data_sq / N) * thresh)) {  // QOPeriods.py:391,418
      stopped_by_test = true;
      break;
    }
    if (nb >= kQoMaxBlocks) {
      status = 3;
      break;
    }
    // ---- strongest gamma-normalised projection of the residual (QOPeriods.py:470-478)
    double best = 0.0;
    int bestp = 0;
    double best_ss = 0.0;  // same lazy comparison as in k_mbest_step1 (gamma norm: ss / p)
    wave_sweep_plan<T, LW>(work, N, geom, plan, wv, n_pass, nw, lane, [&](double ss, int p) {
      if (!(ss > 0.0)) return;
      bool take = bestp == 0;
      if (!take) {
        const double lhs = ss * (double)bestp, rhs = best_ss * (double)p;
        if (lhs > rhs * (1.0 + 1e-14)) {
          take = true;
        } else if (lhs >= rhs * (1.0 - 1e-14)) {
          const double vn = periodic_norm_from_sq(ss, N, p), vb = periodic_norm_from_sq(best_ss, N, bestp);
          take = vn > vb || (vn == vb && p < bestp);
        }
      }
      if (take) {
        best_ss = ss;
        bestp = p;
      }
    });
    best = bestp != 0 ? periodic_norm_from_sq(best_ss, N, bestp) : 0.0;
    if (!(best > 0.0)) bestp = 0;
    wave_argmax(best, bestp);
    if (lane == 0) {
      wbest[wv] = best;
      wbestp[wv] = bestp;
    }
    __syncthreads();
    red_argmax(wbest, wbestp, nw, best, bestp);
    __syncthreads();
    PH_QO_MARK(0)
    if (bestp == 0) break;  // nothing left to explain; the reference keeps looping on zeros
    // ---- rows this period contributes (QOPeriods.py:833-840); a repeated period or one whose
    //      divisors are all present adds none and makes the reference's matrix singular
    int keep = 0;
    for (int k = div_off[bestp]; k < div_off[bestp + 1]; ++k) {
      const int r = div_q[k];
      if (!((seen[r >> 5] >> (r & 31)) & 1u)) keep += phi[r];
    }
    bool repeated = false;
    for (int b = 0; b < nb; ++b) repeated |= (bper[b] == bestp);
    __syncthreads();
    const int row0 = boff[nb];
    if (keep == 0 || repeated) break;  // LinAlgError path (QOPeriods.py:552-559)
    if (row0 + keep > kcap) {
      status = 3;
      break;
    }
    if (tid == 0) {
      for (int k = div_off[bestp]; k < div_off[bestp + 1]; ++k) seen[div_q[k] >> 5] |= 1u << (div_q[k] & 31);
      bper[nb] = bestp;
      bkeep[nb] = keep;
      boff[nb + 1] = row0 + keep;
      bnorm[nb] = best;
    }
    // pair constants of the new block against every block (and itself: unused)
    if (nb < kQoPairTab) {
      for (int e = tid; e < nb; e += blockDim.x) {
        const int pe = bper[e], g = qo_gcd(pe, bestp);
        ptab[2 * (e * kQoPairTab + nb)] = bestp / g;  // row in block e against block nb
        ptab[2 * (e * kQoPairTab + nb) + 1] = pe % bestp;
        ptab[2 * (nb * kQoPairTab + e)] = pe / g;
        ptab[2 * (nb * kQoPairTab + e) + 1] = bestp % pe;
      }
    }
    // right-hand side rows of the new block: fold of the data (QOPeriods.py:782), one wavefront per residue (a
    // small period has few residues with many samples each)
    for (int j = wv; j < keep; j += nw) {
      const int terms = (N - 1 - j) / bestp + 1;
      double sj = 0.0;
      for (int r = lane; r < terms; r += kWave) sj += (double)data[j + (int64_t)r * bestp];
      sj = wave_sum(sj);
      if (lane == 0) rhs[row0 + j] = sj;
    }
    __threadfence_block();
    __syncthreads();
    PH_QO_MARK(1)
    const int K = row0 + keep;
    const int nblk = nb + 1;
    const int KS = K + nblk;  // slots
    // the window buffer is dead from here to the reconstruction: set the solver's vectors up in it
    for (int sl = tid; sl < KS; sl += blockDim.x) {
      int a = 0;
      while (a + 1 < nblk && boff[a + 1] + a + 1 <= sl) ++a;
      const int i = sl - boff[a] - a;
      const bool pad = i >= bkeep[a];  // the zero behind block a
      const int r = boff[a] + i;
      const int terms = pad ? 1 : (N - 1 - i) / bper[a] + 1;
      trm[sl] = terms;
      dv[sl] = pad ? 0.0 : 1.0 / (double)terms;
      rv[sl] = pad ? 0.0 : rhs[r];  // the right-hand side, until the first product turns it into the residual
      wv_[sl] = 0.0;  // (the product never writes the pad slots)
      qv[sl] = 0.0;
      if (pad || r >= row0) xv[sl] = 0.0;  // the rows of earlier steps start from their last weights
    }
    // Work split of the product: a row of block a costs sum_b min(cycle_ab, samples) steps -- the rows of short
    // periods are the long ones.  Block a gets L_a = 1, 2, ..., 16 lanes per row so that no lane walks more than
    // ~1/1024 of all steps; the lanes of a row are neighbours and combine with DPP.
    if (tid == 0) {
      long long total = 0;
      int steps[kQoMaxBlocks];
      for (int a = 0; a < nblk; ++a) {
        const int pa = bper[a], terms = (N - 1) / pa + 1;
        int st = 0;
        for (int b = 0; b < nblk; ++b) {
          if (b == a) continue;
          const int cyc = bper[b] / qo_gcd(pa, bper[b]);
          st += cyc < terms ? cyc : terms;
        }
        steps[a] = st;
        total += (long long)st * bkeep[a];
      }
      // greedy: give the block with the longest walk twice the lanes while all items still fit one round of the
      // workgroup (blocks of periods below 64 always get 16 lanes: their rows have up to N / 2 samples)
      int items = 0;
      for (int a = 0; a < nblk; ++a) {
        blg[a] = bper[a] < 64 ? 4 : 0;
        items += ((bkeep[a] << blg[a]) + 15) & ~15;
      }
      (void)total;
      for (;;) {
        int worst = -1, wst = 24;
        for (int a = 0; a < nblk; ++a)
          if (blg[a] < 4 && (steps[a] >> blg[a]) > wst) {
            wst = steps[a] >> blg[a];
            worst = a;
          }
        if (worst < 0) break;
        const int grown = items - (((bkeep[worst] << blg[worst]) + 15) & ~15) + (((bkeep[worst] << (blg[worst] + 1)) + 15) & ~15);
        if (grown > (int)blockDim.x) break;
        items = grown;
        blg[worst] += 1;
      }
      int off = 0;
      for (int a = 0; a < nblk; ++a) {
        ioff[a] = off;
        off += ((bkeep[a] << blg[a]) + 15) & ~15;  // groups never straddle a 16-lane DPP row
      }
      ioff[nblk] = off;
    }
    __syncthreads();
    // ---- A A^T w = A x by preconditioned conjugate gradients.  The Gram matrix (integer co-occurrence counts,
    //      QOPeriods.py:781) is never formed: its rows are regenerated from the CRT structure inside the product
    //      (qo_offdiag), so the solve touches LDS only -- a dense factorisation of the K x K matrix (K up to ~900,
    //      5 MB) streamed it through HBM K / 32 times.  The diagonal blocks are diagonal (samples per residue); with
    //      that as preconditioner the spectrum is a cluster at 1 plus a few small eigenvalues from the mean /
    //      common-divisor directions the blocks share: 20-70 iterations to 1e-13.
    // With `dots`, the lane that finishes a row also accumulates that row's terms of gamma = (r, z), delta = (w, z)
    // and |r|^2 (vv is z then), so the three sums need no pass of their own.
    double dg = 0.0, dd = 0.0, dr = 0.0;
    auto gram_apply = [&](const double* __restrict__ vv, double* __restrict__ out, bool dots) {
      const int nitems = ioff[nblk];
      for (int v = tid; v < nitems; v += blockDim.x) {
        int a = 0;
        while (a + 1 < nblk && ioff[a + 1] <= v) ++a;
        const int lg = blg[a], L = 1 << lg;
        const int e = v - ioff[a];
        const int i = e >> lg, gl = e & (L - 1);
        const int ka = bkeep[a];
        double acc = 0.0;
        int terms = 1;
        const int sl = boff[a] + a + (i < ka ? i : 0);
        if (i < ka) {  // (the padding items of the last row group only take part in the DPP steps)
          const int pa = bper[a];
          terms = trm[sl];
          for (int b = 0; b < nblk; ++b) {
            if (b == a) continue;
            const int pb = bper[b];
            int cycle, step;
            if (nblk <= kQoPairTab) {
              cycle = ptab[2 * (a * kQoPairTab + b)];
              step = ptab[2 * (a * kQoPairTab + b) + 1];
            } else {
              cycle = pb / qo_gcd(pa, pb);
              step = pa % pb;
            }
            acc += qo_offdiag(vv + boff[b] + b, bkeep[b], pb, cycle, step, i, terms, gl, L);
          }
        }
        if (lg >= 4) acc += dpp_f64<kDppRor8>(acc);
        if (lg >= 3) acc += dpp_f64<kDppHalfMirror>(acc);
        if (lg >= 2) acc += dpp_f64<kDppXor2>(acc);
        if (lg >= 1) acc += dpp_f64<kDppXor1>(acc);
        if (gl == 0 && i < ka) {
          const double z = vv[sl], wrow = fma((double)terms, z, acc);
          out[sl] = wrow;
          if (dots) {
            const double t = rv[sl];
            dg = fma(t, z, dg);
            dd = fma(wrow, z, dd);
            dr = fma(t, t, dr);
          }
        }
      }
    };
    // Single-reduction form (Chronopoulos / Gear): z = D^-1 r, w = G z, and gamma = (r, z), delta = (w, z), |r|^2
    // come out of ONE workgroup reduction per iteration; p and q = G p follow by recurrence.
    bool singular = false;
    {
      gram_apply(xv, qv, false);
      __syncthreads();
      double bb = 0.0;
      for (int r = tid; r < KS; r += blockDim.x) {
        const double bval = rv[r], t = bval - qv[r];
        rv[r] = t;
        zv[r] = t * dv[r];
        bb = fma(bval, bval, bb);
      }
      bb = block_sum(bb, red);  // (its barriers also publish zv)
      // ||r|| <= tol ||A x||: with the condition numbers seen (kappa ~ 3000) the weights are then good to 3e-10 (fp64
      // windows, bar 1e-8) and 3e-6 (float windows, whose residual is stored as float; bar 1e-4)
      const double tol = sizeof(T) == 4 ? 1e-9 : 1e-13;
      const double tol2 = tol * tol * bb;
      const int itmax = 4 * K + 100;
      double gamma_old = 0.0, alpha = 0.0, rr = 1.0 / 0.0;
      int iter = 0;
      for (;; ++iter) {
        dg = dd = dr = 0.0;
        gram_apply(zv, wv_, true);
        // one reduction for the three sums; the partials of odd and even iterations use different slots, so the
        // only barriers of an iteration are the one here and the one behind the vector updates
        double* part = red3 + (iter & 1) * 3 * kMaxWaves;
        const double sg = wave_sum(dg), sd = wave_sum(dd), sr = wave_sum(dr);
        if (lane == 0) {
          part[wv] = sg;
          part[kMaxWaves + wv] = sd;
          part[2 * kMaxWaves + wv] = sr;
        }
        __syncthreads();
        // (all reads in flight before the first addition: the loop over the run-time wave count was 3 x 16 dependent
        // LDS round trips per iteration)
        const double g = uniform_f64(red_combine(part, nw)), d = uniform_f64(red_combine(part + kMaxWaves, nw));
        rr = uniform_f64(red_combine(part + 2 * kMaxWaves, nw));
        if (rr <= tol2 || iter >= itmax) break;
        const double beta = iter == 0 ? 0.0 : g / gamma_old;
        const double denom = iter == 0 ? d : d - beta * g / alpha;
        if (!(denom > 0.0) || !(g > 0.0)) {  // not positive definite: numpy.linalg.solve would raise or return garbage
          singular = true;
          break;
        }
        alpha = g / denom;
        gamma_old = g;
        for (int r = tid; r < KS; r += blockDim.x) {
          const double pn = fma(beta, iter == 0 ? 0.0 : pv[r], zv[r]);
          const double qn = fma(beta, iter == 0 ? 0.0 : qv[r], wv_[r]);
          pv[r] = pn;
          qv[r] = qn;
          xv[r] = fma(alpha, pn, xv[r]);
          const double t = fma(-alpha, qn, rv[r]);
          rv[r] = t;
          zv[r] = t * dv[r];
        }
        __syncthreads();
      }
      if (!(rr <= tol2)) singular = true;  // no convergence (or not finite): a numerically singular dictionary
#ifdef PH_QO_TIMERS
      cg_iters += iter;
#endif
    }
    PH_QO_MARK(2)
    __syncthreads();
    if (singular) {
      // go back one iteration and stop (QOPeriods.py:552-559): the weights return to the last good solve, and the
      // residual -- the solver's vectors may have overwritten the window -- is rebuilt from them below
      for (int r = tid; r < row0; r += blockDim.x) {
        int a = 0;
        while (a + 1 < nb && boff[a + 1] <= r) ++a;
        xv[r + a] = wts[r];
      }
    } else {
      nb += 1;
      for (int r = tid; r < K; r += blockDim.x) {
        int a = 0;
        while (a + 1 < nb && boff[a + 1] <= r) ++a;
        wts[r] = xv[r + a];
      }
    }
    __syncthreads();
    // ---- reconstruction A^T w (QOPeriods.py:795) and the new residual
    double rs = 0.0;
    for (int n = tid; n < N; n += blockDim.x) {
      double rec = 0.0;
      for (int b = 0; b < nb; ++b) {
        const int i = n % bper[b];
        if (i < bkeep[b]) rec += xv[boff[b] + b + i];
      }
      rs += rec * rec;
      work[n] = (T)((double)data[n] - rec);
    }
    if (LW) zero_pad(work, N);
    recon_sq = block_sum(rs, red);
    __syncthreads();
    PH_QO_MARK(3)
    if (singular) break;
  }
  __syncthreads();
#ifdef PH_QO_TIMERS
  if (w < 12 && tid == 0)
    printf("qo timers (100 MHz ticks) sweep %lld rhs %lld cg %lld recon %lld  cg iterations %d K=%d nb=%d\n", tq[0], tq[1], tq[2],
           tq[3], cg_iters, boff[nb], nb);
#endif
  // outputs.  When the loop stopped on the test function the reference reports all periods
  // but the last one, yet keeps the weights / dictionary of all of them (QOPeriods.py:584-592).
  const int n_report = stopped_by_test ? nb - 1 : nb;
  for (int b = tid; b < num; b += blockDim.x) {
    const bool in = b < nb;
    periods_out[w * num + b] = in ? (uint32_t)bper[b] : 0u;
    norms_out[w * num + b] = in ? bnorm[b] : 0.0;
    keeps_out[w * num + b] = in ? bkeep[b] : 0;
  }
  const int Kf = boff[nb];
  for (int r = tid; r < kcap; r += blockDim.x) weights_out[w * (int64_t)kcap + r] = r < Kf ? wts[r] : 0.0;
  for (int n = tid; n < N; n += blockDim.x) resid_out[w * (int64_t)N + n] = work[n];
  if (tid == 0) {
    counts_out[2 * w] = n_report < 0 ? 0 : n_report;
    counts_out[2 * w + 1] = nb;
    status_out[w] = status;
  }
}

// ======================================================================================
// QOPeriods.get_best_period_orthogonal / eq_3 / auto_corr  (QOPeriods.py:1122-1232), the
// Muresan-Parks orthogonal period powers.  One workgroup per window, window and its full
// autocorrelation resident in LDS:
//   r[k]   = sum_{n < N-k} x[n] x[n+k]                                   (auto_corr, :1151-1173)
//   e3[q]  = (q/N) (r[0] + 2 sum_{l=1}^{N//q - 1} r[l q])                 (eq_3, :1122-1149)
//   pows[q] = max(e3[q], 0) - sum_{f | q, f < q} pows[f]                  (:1210-1217)
//           = sum_{d | q} mu(q/d) max(e3[d], 0)   (Moebius inversion of the same recursion),
//   negatives clipped to 0 afterwards, optionally divided by q (:1218-1223).
// ======================================================================================
template <typename T, bool LW>
__global__ __launch_bounds__(kBlockWide) void k_orth_powers(const T* __restrict__ x, int N, int max_p, int normalize,
                                                            const int* __restrict__ mob_off,
                                                            const int* __restrict__ mob_d,
                                                            const int* __restrict__ mob_mu,
                                                            double* __restrict__ gws,
                                                            double* __restrict__ r_out, double* __restrict__ e3_out,
                                                            double* __restrict__ pows_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x;
  // LW: window, autocorrelation and clipped eq. 3 values live in LDS.  Otherwise (long windows) the
  // window is read straight from HBM / L2 and the two work arrays sit in the HBM workspace `gws`.
  const T* xs = x + w * (int64_t)N;
  double *r, *m;
  if constexpr (LW) {
    Carve cv(smem);
    T* stage = cv.take<T>(N);
    r = cv.take<double>(N);
    m = cv.take<double>(max_p);
    load_window(xs, stage, N);
    __syncthreads();
    xs = stage;
  } else {
    r = gws + w * ((int64_t)N + max_p);
    m = r + N;
  }
  // autocorrelation: lags k and N-1-k are paired on one thread (N-k plus k+1 products = N+1)
  for (int k = tid; k < (N + 1) / 2; k += blockDim.x) {
    const int k2 = N - 1 - k;
    double a = 0.0, b = 0.0;
    for (int n = 0; n + k < N; ++n) a += (double)xs[n] * (double)xs[n + k];
    if (k2 != k)
      for (int n = 0; n + k2 < N; ++n) b += (double)xs[n] * (double)xs[n + k2];
    r[k] = a;
    if (k2 != k) r[k2] = b;
  }
  __threadfence_block();
  __syncthreads();
  if (r_out)
    for (int k = tid; k < N; k += blockDim.x) r_out[w * (int64_t)N + k] = r[k];
  for (int q = tid; q < max_p; q += blockDim.x) {
    double v = 0.0;
    if (q >= 1) {
      const int M = N / q;
      double second = 0.0;
      for (int l = 1; l < M; ++l) second += r[l * q];
      v = ((double)q / (double)N) * (r[0] + 2.0 * second);
    }
    if (e3_out) e3_out[w * (int64_t)max_p + q] = v;
    m[q] = fmax(v, 0.0);
  }
  __threadfence_block();
  __syncthreads();
  for (int q = tid; q < max_p; q += blockDim.x) {
    double v = 0.0;
    for (int k = mob_off[q]; k < mob_off[q + 1]; ++k) v += (double)mob_mu[k] * m[mob_d[k]];
    v = v < 0.0 ? 0.0 : v;
    if (normalize && q >= 1) v = v / (double)q;
    pows_out[w * (int64_t)max_p + q] = q >= 1 ? v : 0.0;
  }
}

// ======================================================================================
// Periods.periodic_norm over a batch (Periods.py:221-241); streams from HBM, any N.
// ======================================================================================
template <typename T>
__global__ __launch_bounds__(kBlock) void k_periodic_norm(const T* __restrict__ x, int N, int p_div,
                                                          double* __restrict__ out) {
  __shared__ double red[kRedDoubles];
  const int64_t w = blockIdx.x;
  const T* row = x + w * (int64_t)N;
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const double t = (double)row[n];
    acc += t * t;
  }
  const double ss = block_sum(acc, red);
  if (threadIdx.x == 0) out[w] = periodic_norm_from_sq(ss, N, p_div);
}

// ======================================================================================
// RamanujanPeriods.project(x, basis) (RamanujanPeriods.py:124-131) for an arbitrary
// dictionary: row <- row / max(row); out[i] = float32(dot(x, row) * row).  One workgroup
// per dictionary row, rows streamed from HBM twice (max, then dot+scale from L2).
// ======================================================================================
__global__ __launch_bounds__(kBlock) void k_dict_project(const double* __restrict__ x,
                                                         const double* __restrict__ basis, int N,
                                                         float* __restrict__ out) {
  __shared__ double red[kRedDoubles];
  const int64_t i = blockIdx.x;
  const double* row = basis + i * (int64_t)N;
  double m = -INFINITY;
  for (int n = threadIdx.x; n < N; n += blockDim.x) m = fmax(m, row[n]);
  m = wave_max(m);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = fmax(m, red[k]);
  __syncthreads();
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) acc += x[n] * (row[n] / m);
  const double dot = block_sum(acc, red);
  for (int n = threadIdx.x; n < N; n += blockDim.x) out[i * (int64_t)N + n] = (float)(dot * (row[n] / m));
}

}  // namespace ph
