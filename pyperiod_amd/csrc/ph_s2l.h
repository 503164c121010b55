// Periods.small_to_large (Periods.py:246-287), window-pair screen with a speculative in-order period QUEUE and an
// fp64 staging buffer in LDS (round 4; gfx950 / CDNA4 only).  Included from ph_kernels.h.
//
//   Two fp64 windows share a workgroup and the pair window of ph_pair.h: the ascending screen of k_small_to_large
//   (||r||^2 - ||P_q r||^2 as an estimate of the reference's norm drop, Periods.py:274-281) runs on the float images of
//   both -- one pass of the wave-per-period fold per candidate period serves the two windows.  A period whose estimate,
//   widened by the rigorous float radius (pair_radius) and the fp64 terms of k_small_to_large's bound, reaches the
//   threshold is evaluated exactly on the fp64 residual (row-order means, direct sum of squares of the trial residual),
//   so accept decisions, powers and bases are those of the one-window kernel, bit for bit.
//
//   * 16 wavefronts per pair, two workgroups per CU.  A workgroup lasts half as long as an 8-wave one, so a launch
//     ends with half the tail, and the LDS beside the pair window holds ONE fp64 staging buffer: the residual of the
//     window whose candidate is being evaluated.  It is a write-back cache of the HBM workspace -- an event reads the
//     residual through L2 only when the other window owned the buffer and writes it back only when it is evicted dirty
//     -- and the exact evaluation and the update (Periods.py:274-286) run from LDS like the one-window kernel's.
//   * No barrier per batch of periods.  The wavefronts draw candidate periods in ascending order from a ticket
//     counter in LDS; each tests its own screen value and a flagged period lowers `limit`, after which no larger
//     ticket is started.  The workgroup meets at a barrier only when the tickets up to the limit are done: about once
//     per accepted period instead of once per 32 candidates, and the wavefronts never wait for each other in between.
//     A flag of window w at q is its FIRST flag iff every ticket below q was executed, i.e.
//     q < min(first skipped ticket, next unclaimed ticket); a later flag is dropped and found again.
//   * The flag test of a pass is a float compare.  With the state of window w (rsq = ||r||^2, A, c = rsq / A, scale s:
//     see k_small_to_large) the fp64 test flags q iff  v > s^2 (rsq - (A + kappa_q c)^2 (1 + 1e-12)).  Here
//     T0 - T1 kappa_q with T0 <= s^2 (rsq - A^2 (1 + 1e-12)), T1 >= s^2 (2 A c + c^2 kappa_max)(1 + 1e-12) and kappa_q
//     from a host table (rounded up) is a lower bound of that threshold because kappa_q <= kappa_max = kappa_2: every
//     period the fp64 test would flag is flagged, and every flagged period is decided by the exact evaluation.
//   The exact phases use the thread mapping of the 512-thread kernels (threads >= kBlockWide only help with the
//   staging copies), which keeps every sum in the order of k_small_to_large<double>.
#pragma once

namespace ph {

// Exact evaluation of a candidate period p <= kBlockWide (Periods.py:274-281), shared by the one-window and the
// window-pair kernel so that both sum in the same order.  A short period has few residues with hundreds of rows each:
// one thread per residue walking its column (the order of Periods.project) left all but p threads idle behind a chain
// of N/p dependent additions -- 3 us per event, a quarter of the kernel.  Here the means come from split_row_means
// (ph_device.h: the rows of a residue dealt to up to 64 threads, partial sums combined in a fixed order), and the
// trial residual's sum of squares and the update run FLAT over the samples (thread t takes
// n = t, t + width, ... with the mean index kept incrementally).  The mean of a residue is then a differently
// associated sum than np.sum(cp, 0) (Periods.py:194) -- within 1e-15 of it; small_to_large's bases and powers carry a
// 1e-10 bar, only Periods.project itself (ph_project_batch) is held to bit-identity -- and np.linalg.norm has no
// defined order anyway.  Threads >= width do nothing; msm holds >= width elements.
// `part` (>= width elements) takes the partial sums of split_row_means; it may be `msm` itself (one more barrier then).
constexpr int kFlatBatch = 8;  // samples a thread takes per trip of the flat loops (N = 4096, 512 threads: one trip)
struct S2LNoMark {
  __device__ __forceinline__ void operator()(int) const {}
};
template <typename T, typename MK = S2LNoMark>
__device__ __forceinline__ double s2l_flat_trial(const T* __restrict__ work, T* msm, T* part, int N, int p, int tid, int width,
                                                 MK&& mark = MK()) {
  split_row_means(work, msm, part, N, p, tid, width);
  mark(4);
  double tsq = 0.0;
  if (tid < width) {
    // kFlatBatch samples per trip with all their LDS reads in flight before the first is used (a rolled loop paid one
    // LDS round trip per sample: 8 of them at N = 4096); the additions keep the order n = tid, tid + width, ...
    int idx = tid % p;
    const int step = width % p;
    for (int n0 = tid; n0 < N; n0 += kFlatBatch * width) {
      T x[kFlatBatch], m[kFlatBatch];
#pragma unroll
      for (int k = 0; k < kFlatBatch; ++k) {
        const int n = n0 + k * width;
        x[k] = work[n < N ? n : n0];
        m[k] = msm[idx];
        idx += step;
        idx = idx >= p ? idx - p : idx;
      }
#pragma unroll
      for (int k = 0; k < kFlatBatch; ++k) {
        const double t = (double)(x[k] - m[k]);
        tsq = n0 + k * width < N ? fma(t, t, tsq) : tsq;
      }
    }
  }
  return tsq;
}

// residual <- residual - projection for the means in `msm`; brow (optional) receives the projection, extra(n, v) sees
// every new residual sample
template <typename T, typename F>
__device__ __forceinline__ void s2l_flat_update(T* __restrict__ work, const T* __restrict__ msm, int N, int p, int tid, int width,
                                                T* __restrict__ brow, F&& extra) {
  if (tid >= width) return;
  int idx = tid % p;
  const int step = width % p;
  for (int n0 = tid; n0 < N; n0 += kFlatBatch * width) {  // batched like s2l_flat_trial
    T x[kFlatBatch], m[kFlatBatch];
#pragma unroll
    for (int k = 0; k < kFlatBatch; ++k) {
      const int n = n0 + k * width;
      x[k] = work[n < N ? n : n0];
      m[k] = msm[idx];
      idx += step;
      idx = idx >= p ? idx - p : idx;
    }
#pragma unroll
    for (int k = 0; k < kFlatBatch; ++k) {
      const int n = n0 + k * width;
      if (n < N) {
        const T v = x[k] - m[k];
        if (brow) brow[n] = m[k];
        work[n] = v;
        extra(n, v);
      }
    }
  }
}

#ifdef PH_CLOCKS
constexpr int kStampCap = 65536;
__device__ long long g_ph_stamps[4 * kStampCap];  // diagnostic build only (tools/s2l_clocks.py)
#endif
constexpr int kS2LInf = 0x7fffffff;

enum { S2L_TICK = 0, S2L_LIMIT, S2L_STOP0, S2L_STOP1, S2L_SKIP, S2L_QWORDS = 8 };

// LDS of one workgroup (host and device agree through this one function)
__host__ __device__ inline size_t s2l_pair_lds_bytes(int N) {
  return carve_bytes(N + kPad, 8) + carve_bytes(N, 8) + carve_bytes(kRedDoubles, 8) + carve_bytes(kBlockWide / 2, 8) + carve_bytes(kBlockWide, 8) + carve_bytes(8, 8) +
         carve_bytes(4, 4) + carve_bytes(4, 4) + carve_bytes(S2L_QWORDS, 4);
}

// kappa_q of the flag test: float radius of the screen + the fp64 terms of k_small_to_large's bound
__host__ __device__ inline double s2l_kappa(int N, int rows, int q) {
  return 1.5 * (2.0 * (double)rows + (double)(q >> 6) + 32.0) * 5.9604644775390625e-08 * (1.0 + 1e-9) +
         ((double)N / 256.0 + 32.0) * 2.220446049250313e-16;
}

// float image of window w of a pair times a power of two (its scale is renewed when the residual has collapsed)
__device__ __forceinline__ void s2l_pair_rescale(float* __restrict__ pwf, int w, int N, float up) {
  for (int n = threadIdx.x; n < N; n += blockDim.x) pwf[2 * n + w] *= up;
}

// staging copies by the whole workgroup (16-byte vectors when both rows are 16-byte aligned)
__device__ __forceinline__ void s2l_copy(const double* __restrict__ src, double* __restrict__ dst, int N) {
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && (N & 1) == 0) {
    const double2* s = reinterpret_cast<const double2*>(src);
    double2* d = reinterpret_cast<double2*>(dst);
    const int nv = N / 2, bd = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < nv; i0 += 4 * bd) {  // four loads in flight per thread: one L2 round trip at N = 4096
      double2 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = s[i0 + k * bd < nv ? i0 + k * bd : i0];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i0 + k * bd < nv) d[i0 + k * bd] = v[k];
    }
  } else {
    for (int i = threadIdx.x; i < N; i += blockDim.x) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_small_to_large_pair(
    const double* __restrict__ x, int W, int N, double thresh, int n_periods, const PGeomF* __restrict__ geomf,
    const float* __restrict__ kapf, double* __restrict__ gres, int cap, int* __restrict__ counts,
    int* __restrict__ periods_out, double* __restrict__ powers_out, double* __restrict__ bases_out,
    int* __restrict__ status_out, int* __restrict__ max_count, int* __restrict__ next_pair) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Carve cv(smem);
  f2* pw = cv.take<f2>(N + kPad);
  double* stg = cv.take<double>(N);
  double* red = cv.take<double>(kRedDoubles);
  double* msm = cv.take<double>(kBlockWide / 2);  // means of a candidate period dealt to several threads (<= kBlockWide / 2)
  double* prt = cv.take<double>(kBlockWide);  // ... its partial sums; the means of a period in (kBlockWide / 2, kBlockWide]
  // per window w: st[w] ||residual||^2, st[2+w] periodic_norm(residual), st[4+w] periodic_norm(data), st[6+w] scale of
  // the float image; thf[2w], thf[2w+1]: T0, T1 of the flag test; ct[w] periods accepted
  double* st = cv.take<double>(8);
  float* thf = cv.take<float>(4);
  int* ct = cv.take<int>(4);
  int* qc = cv.take<int>(S2L_QWORDS);

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  constexpr int kEx = kBlockWide;  // threads of the exact phases: the mapping of k_small_to_large (bit-identical sums)
  const bool ex = tid < kEx;
  const int bk = (int)blockDim.x > kEx ? kEx : 0;  // bookkeeping thread of the accept path
  const size_t gstride = win_stride((size_t)N);
  float* pwf = reinterpret_cast<float*>(pw);
  const double sqrtN = uniform_f64(sqrt((double)N));
  const double kmax = (double)kapf[2];  // kappa is largest for the shortest period (most rows)

  // thresholds of the flag test from the state of window w (thread 0)
  auto set_thresholds = [&](int w, double rsq, double rn, double dn, double sc) {
    const double A = (rn - (thresh - 1e-13) * dn) * sqrtN;
    const double c = rsq / A, s2 = sc * sc, up = 1.0 + 1e-12;
    double T0, T1;
    if (A > 0.0) {
      T0 = (rsq - A * A * up) * s2;
      T1 = (2.0 * A * c + c * c * kmax) * up * s2;
    } else {  // the residual is already below thresh x data: the fp64 test uses min(kappa rsq, (kappa c)^2) <= kappa_max rsq
      T0 = rsq * (1.0 - kmax * up) * s2;
      T1 = 0.0;
    }
    T0 -= 9.5367431640625e-07 * (fabs(T0) + T1 * kmax);  // 2^-20: the roundings to float and of the float fma
    float t0 = (float)T0, t1 = (float)T1;
    if ((double)t0 > T0) t0 = nextafterf(t0, -INFINITY);
    if ((double)t1 < T1) t1 = nextafterf(t1, INFINITY);
    thf[2 * w] = t0;  // NaN (non-finite windows): every comparison fails -> every period is evaluated exactly
    thf[2 * w + 1] = t1;
  };

  for (int i = tid; i < kPad; i += blockDim.x) pw[N + i] = f2_zero();
  // Persistent workgroups: the grid fills the chip once and every workgroup draws pairs from a device counter until
  // the batch is done (a fresh workgroup per pair left ~5 % of the slots empty between an exit and the next dispatch).
  const int64_t npairs = ((int64_t)W + 1) / 2;
  for (int64_t pair = blockIdx.x; pair < npairs;) {
  int pos0 = n_periods + 1, pos1 = n_periods + 1;  // positions of the two windows: identical in every thread
  for (int w = 0; w < 2; ++w) {
    const int64_t gw = 2 * pair + w;
    const bool exists = gw < W;
    __syncthreads();
    if (exists) s2l_copy(x + gw * (int64_t)N, stg, N);
    __syncthreads();
    double acc = 0.0;
    if (exists && ex)
      for (int n = tid; n < N; n += kEx) {
        const double v = stg[n];
        acc = fma(v, v, acc);
      }
    const double rsq = block_sum(acc, red);
    const double sc = uniform_f64(pair_pick_scale(rsq, N));
    for (int n = tid; n < N; n += blockDim.x) pwf[2 * n + w] = exists ? (float)(stg[n] * sc) : 0.0f;
    if (tid == 0) {
      const double dnorm = sqrt(rsq) / sqrtN;  // data_norm, Periods.py:269
      st[w] = rsq;
      st[2 + w] = st[4 + w] = dnorm;
      st[6 + w] = sc;
      set_thresholds(w, rsq, dnorm, dnorm, sc);
      ct[w] = 0;
    }
    if (exists) (w ? pos1 : pos0) = 2;
  }
  // the staging buffer now holds the input of window 1 (if it exists), unchanged
  int own = (2 * pair + 1 < W) ? 1 : -1;
  bool dirty = false, moved0 = false, moved1 = false;  // moved: the residual is no longer the input
#ifdef PH_S2L_TIMERS
  long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tp0 = wall_clock64();
  int nev = 0, nepoch = 0, nswap = 0;
  if (tid == 0) qc[5] = 0;
#define PH_S2LQ_MARK(k)                    \
  {                                        \
    const long long now_ = wall_clock64(); \
    tp[k] += now_ - tp0;                   \
    tp0 = now_;                            \
  }
#else
#define PH_S2LQ_MARK(k)
#endif
#ifdef PH_CLOCKS
  const long long wk0 = wall_clock64();
#endif

  bool had_event = false;  // the epoch before this one held a barrier after its control words were read
  for (;;) {
    const int start = min(pos0, pos1);
    if (start > n_periods) break;
    if (!had_event) __syncthreads();  // the previous epoch's control words have been read
    if (tid == bk) {
      qc[S2L_TICK] = start + nw;
      qc[S2L_LIMIT] = n_periods;
      qc[S2L_STOP0] = qc[S2L_STOP1] = qc[S2L_SKIP] = kS2LInf;
    }
    __syncthreads();  // ... and: the updates of the events (residual, float image, thresholds) are complete
    had_event = false;
    prio_long_phase();
    {
      const float t00 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(thf[0]))),
                  t01 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(thf[1]))),
                  t10 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(thf[2]))),
                  t11 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(thf[3])));
      int q = start + wv;
      for (;;) {
        const int lim = __builtin_amdgcn_readfirstlane(*(lds_int_ptr)&qc[S2L_LIMIT]);  // (a plain `volatile int*` became a flat load)
        if (q > lim) {
          if (q <= n_periods && pair_lane() == 0) atomicMin(&qc[S2L_SKIP], q);
          break;
        }
        int nxt = 0;
        if (pair_lane() == 0) nxt = lds_ticket(&qc[S2L_TICK]);  // the next ticket: claimed now, looked at after this pass
        const float kap = kapf[q];
        const PGeomF gq = geomf[q];
        // packed totals: lanes 0-31 hold the value of window 0, lanes 32-63 that of window 1 -- each half tests its own
        const float z = pair_reduce1<false>(q >= 64 ? pair_pass_single(pw, q, gq) : pair_partial_small(pw, N, q, gq));
        {
          const int l = pair_lane();
          const bool second = l >= 32;
          const float bound = second ? fmaf(-t11, kap, t10) : fmaf(-t01, kap, t00);
          const bool flag = q >= (second ? pos1 : pos0) && !(z <= bound);  // NaN -> evaluate
          if (flag && (l & 31) == 0) {
            atomicMin(&qc[second ? S2L_STOP1 : S2L_STOP0], q);
            atomicMin(&qc[S2L_LIMIT], q);
          }
        }
        q = lds_ticket_value(nxt);
#ifdef PH_S2L_EXTRA_SALU  // sensitivity probe: PH_S2L_EXTRA_SALU x 8 scalar adds (or vector adds with PH_S2L_EXTRA_VALU) per pass
        {
          int dummy_s = q;
#pragma unroll
          for (int e = 0; e < PH_S2L_EXTRA_SALU; ++e)
            asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7" : "+s"(dummy_s) : : "scc");
          if (dummy_s == 0x7fffff01) qc[7] = dummy_s;
        }
#endif
#ifdef PH_S2L_EXTRA_VALU
        {
          int dummy_v = pair_lane();
#pragma unroll
          for (int e = 0; e < PH_S2L_EXTRA_VALU; ++e)
            asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 3, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 5, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, 7, %0" : "+v"(dummy_v));
          if (dummy_v == 0x7fffff01) qc[7] = dummy_v;
        }
#endif
#ifdef PH_S2L_TIMERS
        if (pair_lane() == 0) atomicAdd(&qc[5], 1);  // passes executed
#endif
      }
    }
    __syncthreads();
    PH_S2LQ_MARK(0)
#ifdef PH_S2L_TIMERS
    nepoch += 1;
#endif
    prio_short_phase();
    // every ticket below `front` was executed; a flag at or beyond it is not known to be the window's first
    // (the events below run beside the other workgroup's screen, whose folds keep the LDS queue of the CU full: an LDS
    // round trip costs several hundred cycles here, so everything an event needs from LDS is read in this one batch)
    // Everything read here is identical in all lanes and goes to SCALAR registers at once: as vector registers these
    // values (and the positions, owner and flags derived from them) were spilled around the screen loop, and every
    // reload is a scratch access through L2 -- a dozen of them sat on the critical path of an event.
    const int front = __builtin_amdgcn_readfirstlane(min(qc[S2L_SKIP], qc[S2L_TICK]));
    int c0 = __builtin_amdgcn_readfirstlane(qc[S2L_STOP0]), c1 = __builtin_amdgcn_readfirstlane(qc[S2L_STOP1]);
    const double st_rn0 = uniform_f64(st[2]), st_rn1 = uniform_f64(st[3]), st_dn0 = uniform_f64(st[4]), st_dn1 = uniform_f64(st[5]),
                 st_sc0 = uniform_f64(st[6]), st_sc1 = uniform_f64(st[7]);
    const int st_cnt0 = __builtin_amdgcn_readfirstlane(ct[0]), st_cnt1 = __builtin_amdgcn_readfirstlane(ct[1]);
    if (c0 >= front) c0 = kS2LInf;
    if (c1 >= front) c1 = kS2LInf;
    // ---- exact evaluation of each window's candidate (Periods.py:274-286); the owner of the staging buffer first
    for (int k = 0; k < 2; ++k) {
      const int w = (own == 1) ? 1 - k : k;
      const int cand = w ? c1 : c0;
      if (cand == kS2LInf) {  // no candidate: everything below `front` is decided for this window
        if (w)
          pos1 = max(pos1, front);
        else
          pos0 = max(pos0, front);
        continue;
      }
      const int64_t gw = 2 * pair + w;
      if (own != w) {  // swap the staging buffer: write the other window's residual back if it has changed
        if (own >= 0 && dirty) {
          __syncthreads();
          s2l_copy(stg, gres + (2 * pair + own) * gstride, N);
          __threadfence_block();
        }
        __syncthreads();
        s2l_copy((w ? moved1 : moved0) ? gres + gw * gstride : x + gw * (int64_t)N, stg, N);
        own = w;
        dirty = false;
#ifdef PH_S2L_TIMERS
        nswap += 1;
#endif
        __syncthreads();
      }
      had_event = true;
      const double rn = w ? st_rn1 : st_rn0, dn = w ? st_dn1 : st_dn0, sc = w ? st_sc1 : st_sc0;
      const int count = w ? st_cnt1 : st_cnt0;
      const Fold f(N, cand);
      double tsq = 0.0;
      const bool flat = cand <= kEx;  // means through LDS, sums and update flat over the samples (s2l_flat_trial)
      if (flat) {
#ifdef PH_S2L_TIMERS
        PH_S2LQ_MARK(7)  // swap + entry
        tsq = s2l_flat_trial(stg, cand <= kEx / 2 ? msm : prt, prt, N, cand, tid, kEx, [&](int k) { PH_S2LQ_MARK(k) });
        PH_S2LQ_MARK(5)  // flat trial loop
#else
        tsq = s2l_flat_trial(stg, cand <= kEx / 2 ? msm : prt, prt, N, cand, tid, kEx);
#endif
      } else if (ex) {
        for (int j = tid; j < cand; j += kEx) {
          const double m = residue_mean(stg, f, j, false);
          const int cnt = f.count(j);
          for (int r = 0; r < cnt; ++r) {
            const double t = stg[r * cand + j] - m;
            tsq = fma(t, t, tsq);
          }
        }
      }
      tsq = block_sum_once(tsq, red);  // `red` rests until the next event: barriers in between
      PH_S2LQ_MARK(1)
#ifdef PH_S2L_TIMERS
      nev += 1;
#endif
      const double tn = uniform_f64(sqrt(tsq) / sqrtN);
      const double imposed = uniform_f64((rn - tn) / dn);
      if (imposed > thresh) {  // strict, Periods.py:281
        double* brow = (bases_out && count < cap) ? bases_out + (gw * cap + count) * (int64_t)N : nullptr;
        if (flat) {
          s2l_flat_update(stg, cand <= kEx / 2 ? msm : prt, N, cand, tid, kEx, brow, [&](int n, double v) { pwf[2 * n + w] = (float)(v * sc); });
        } else if (ex)
          for (int j = tid; j < cand; j += kEx) {
            const double m = residue_mean(stg, f, j, false);
            const int cnt = f.count(j);
            for (int r = 0; r < cnt; ++r) {
              const int n = r * cand + j;
              const double v = stg[n] - m;
              if (brow) brow[n] = m;
              stg[n] = v;
              pwf[2 * n + w] = (float)(v * sc);
            }
          }
        dirty = true;
        if (w)
          moved1 = true;
        else
          moved0 = true;
        double sc_now = sc;
        if (pair_usable(tsq) && tsq * sc * sc < 9.0e-13 * (double)N) {  // float image below 2^-20 RMS: renew its scale
          __syncthreads();
          const double sc2 = uniform_f64(pair_pick_scale(tsq, N));
          s2l_pair_rescale(pwf, w, N, (float)(sc2 / sc));
          sc_now = sc2;
        }
        if (tid == bk) {  // a thread that has no part in the update (it runs meanwhile)
          st[6 + w] = sc_now;
          set_thresholds(w, tsq, tn, dn, sc_now);
          if (count < cap) {
            periods_out[gw * cap + count] = cand;
            powers_out[gw * cap + count] = imposed;
          }
          ct[w] = count + 1;
          st[w] = tsq;
          st[2 + w] = tn;
        }
      }
      if (w)
        pos1 = cand + 1;
      else
        pos0 = cand + 1;
      PH_S2LQ_MARK(2)
    }
  }
#ifdef PH_S2L_TIMERS
  if (pair < 6 && tid == 0)
    printf("s2l queue timers (100 MHz ticks) screen %lld exact(reduce+decide) %lld update %lld swap+entry %lld partials %lld means %lld trial %lld  epochs %d events %d swaps %d passes %d accepts %d %d\n",
           tp[0], tp[1], tp[2], tp[7], tp[3], tp[4], tp[5], nepoch, nev, nswap, qc[5], ct[0], ct[1]);
#endif
#ifdef PH_CLOCKS
  if (tid == 0 && pair < kStampCap) {  // start, end (100 MHz), hardware id, accepts: read back by ph_debug_stamps
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* o = g_ph_stamps + 4 * (size_t)pair;
    o[0] = wk0;
    o[1] = wall_clock64();
    o[2] = ((long long)(xcc & 15) << 32) | hwid;
    o[3] = ct[0] + ct[1];
  }
#endif
  __syncthreads();
  if (tid < 2) {
    const int64_t gw = 2 * pair + tid;
    if (gw < W) {
      const int count = ct[tid];
      counts[gw] = count;
      status_out[gw] = count > cap ? 3 : 0;
      if (count > cap) atomicMax(max_count, count);  // rare: the host retries with this capacity
    }
  }
  if (tid == 0) qc[6] = (int)gridDim.x + atomicAdd(next_pair, 1);  // the host zeroes the counter before the launch
  __syncthreads();
  pair = __builtin_amdgcn_readfirstlane(qc[6]);
  }  // pairs
}

}  // namespace ph
