/*
 * periodhip.h -- C ABI of libperiod_hip.so: the MI355X (gfx950) implementation of the
 * pyPeriod projection hot path.
 *
 * The reference (woolgathering/pyPeriod v1) has no FFI; its boundary for this path is the
 * Python class surface exported at pyPeriod/__init__.py:1-3.  Each entry point below is what
 * a binding for that surface needs; the reference code it replaces is cited per function
 * (file:line into /root/reference/pyPeriod/).  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / numpy types.
 *   - every function returns an int status: PH_OK (0) or a negative PH_E_* code; the text of
 *     the last failure on the calling thread is returned by ph_last_error().
 *   - windows are row-major (W, N) contiguous, dtype PH_F64 or PH_F32 (W = number of
 *     independent signal windows, N = samples per window).  One window == one call of the
 *     reference.
 *   - array arguments are HOST pointers by default: the library stages them through device
 *     buffers owned by the context and the call is synchronous.  With PH_FLAG_DEVICE in
 *     `flags` all *array* arguments (x and every output) are DEVICE pointers on the
 *     context's device, nothing is copied, and the call only enqueues work on the context's
 *     stream (use ph_sync / ph_timer_*).  The small integer tables (p_list, orth_*, fac_*)
 *     are always host pointers.
 *   - the caller owns every buffer it passes; the library never frees or retains them.
 *   - a ph_ctx is bound to one device and is not re-entrant; use one context per thread/GPU.
 *   - set-order tables: two places of the reference iterate a CPython `set` of divisors
 *     (Periods.py:209 and Periods.py:549) and the result depends on that order.  The caller
 *     supplies the order as dense CSR tables indexed by period: entries for period p are
 *     q[off[p]] .. q[off[p+1]-1], off has (table_max_p + 2) entries.  The Python host builds
 *     them from real CPython sets (pyperiod_amd/_factors.py).
 */
#ifndef PERIODHIP_H
#define PERIODHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PH_VERSION 100 /* 0.1.0 */

/* status codes */
#define PH_OK 0
#define PH_E_ARG (-1)         /* bad argument (shape, range, NULL, unsupported N) */
#define PH_E_HIP (-2)         /* a HIP runtime call failed */
#define PH_E_NOMEM (-3)       /* device or host allocation failed */
#define PH_E_CAP (-4)         /* output capacity too small; see the function's doc */
#define PH_E_UNSUPPORTED (-5) /* valid request this build does not implement */

/* dtype of the window data */
#define PH_F64 0
#define PH_F32 1

/* flags */
#define PH_FLAG_TRUNC 1u  /* trunc_to_integer_multiple (Periods.py:178-184) */
#define PH_FLAG_ORTH 2u   /* orthogonalize (Periods.py:208-214); needs orth tables */
#define PH_FLAG_SINGLE 4u /* return_single_period (Periods.py:216-217): only out[..., :p] written */
#define PH_FLAG_DEVICE 8u /* array arguments are device pointers, call is asynchronous */
#define PH_FLAG_NOSYNC 16u /* with PH_FLAG_DEVICE: never synchronise, not even to report PH_E_CAP */

/* sweep modes */
#define PH_SWEEP_NORM 0       /* periodic_norm(project(x,p))        Periods.py:507-508 */
#define PH_SWEEP_NORM_GAMMA 1 /* periodic_norm(project(x,p), p)     Periods.py:509-510 */
#define PH_SWEEP_MAXABS 2     /* max_s |sum(x[s::p])|               Periods.py:327-331 */

/* per-window status words written by the algorithm kernels */
#define PH_ST_OK 0
#define PH_ST_NO_PERIOD 1 /* no candidate period had a positive norm (reference raises) */
#define PH_ST_ITER_CAP 2  /* iteration bound hit before `num` periods were found */
#define PH_ST_CAP 3       /* more accepted periods than `cap` (small_to_large) */

typedef struct ph_ctx ph_ctx;

/* ---- library / context ---------------------------------------------------------------- */
int ph_version(void);
const char* ph_last_error(void);
int ph_device_count(int* count);
/* Create a context on HIP device `device` with its own non-blocking stream. */
int ph_create(int device, ph_ctx** out);
int ph_destroy(ph_ctx* ctx);
/* Borrow an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL
 * restores the context's own (non-blocking) stream.  The device's default stream has the
 * handle 0, which cannot be told from NULL: pass PH_STREAM_DEFAULT for it -- work the caller
 * enqueued on the default stream (torch's current stream unless changed) is then ordered with
 * the library's kernels. */
#define PH_STREAM_DEFAULT ((void*)1)
int ph_set_stream(ph_ctx* ctx, void* hip_stream);
int ph_sync(ph_ctx* ctx);
/* HIP-event timer on the context's stream (the stream the kernels run on). */
int ph_timer_begin(ph_ctx* ctx);
int ph_timer_end(ph_ctx* ctx, float* elapsed_ms);
/* Per-kernel timing: while enabled, every kernel the library launches on this context is
 * bracketed with HIP events on the context's stream (up to 256 launches since the last
 * ph_profile_enable).  ph_profile_read synchronises the stream and returns the elapsed
 * milliseconds of launch i in ms[i]; ph_profile_name gives the kernel's name. */
int ph_profile_enable(ph_ctx* ctx, int on);
int ph_profile_read(ph_ctx* ctx, float* ms, int cap, int* count);
const char* ph_profile_name(ph_ctx* ctx, int i);
/* Multiprocessor count and per-workgroup LDS limit of the context's device. */
int ph_device_info(ph_ctx* ctx, int* num_cu, int* lds_bytes);
/* Largest N for which a window of `dtype` stays LDS-resident (the fast path).  Longer windows are
 * accepted by every entry point: the window then lives in (or is read straight from) HBM / L2 --
 * project, sweep, m_best, small_to_large, best_correlation, ramanujan_norms, best_frequency,
 * qo_find_periods, fold_sums and orth_powers alike.  `flags` is ignored. */
int ph_max_window(ph_ctx* ctx, int dtype, unsigned flags, int* max_n);

/* Pass plan of the norm sweeps (ph_sweep norm modes, ph_m_best, ph_qo_find_periods) over the
 * candidate periods [p_lo, p_hi]: n_periods = p_hi - p_lo + 1 periods are produced by n_pass
 * passes over the LDS-resident window (a pass at base period p also yields the folds of 2p
 * and 4p).  n_pass * N * sizeof(T) is the number of bytes one sweep reads from LDS. */
int ph_sweep_plan_info(ph_ctx* ctx, int p_lo, int p_hi, int* n_pass, int* n_periods);

/* Measurement helper: which step-1 kernel ph_m_best runs for these arguments.  fp64 windows in plain mode
 * that fit the LDS twice are screened two windows per workgroup in packed float (windows_per_workgroup = 2,
 * one 8-byte LDS element carries a sample of both windows, lds_bytes_per_sample = 8 per PAIR) and only the
 * survivors of the screen are re-evaluated in fp64; otherwise one window per workgroup is folded in its
 * own precision (1, sizeof(T)).  A fold pass reads N * lds_bytes_per_sample bytes from LDS per workgroup. */
int ph_m_best_info(ph_ctx* ctx, int dtype, int N, int num, int min_length, int max_length, unsigned flags,
                   int* windows_per_workgroup, int* lds_bytes_per_sample);

/* Measurement helper: the pass plan that step-1 kernel walks per sweep over [min_length, max_length] (max_length < 0:
 * N / 3, Periods.py:486): n_pass passes over the window for n_periods candidate periods.  The window-pair kernel
 * takes the periods up to 64 in chains (one row-split pass at L yields L, L/2, L/4, ...), so its plan is shorter
 * than ph_sweep_plan_info's. */
int ph_m_best_plan_info(ph_ctx* ctx, int dtype, int N, int num, int min_length, int max_length, unsigned flags,
                        int* n_pass, int* n_periods);

/* ---- Periods.periodic_norm over a batch (Periods.py:221-241) ---------------------------
 * out[w] = ||x[w]||_2 / sqrt(N), additionally / sqrt(p) when p > 0.  Any N. */
int ph_periodic_norm(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int p,
                     unsigned flags, double* out);

/* ---- K1: Periods.project over a batch (Periods.py:142-219) ------------------------------
 * out[w, k, :] = project(x[w], p_list[k], trunc, orth)      shape (W, n_p, N), dtype of x.
 * Non-orth results are bit-identical to the reference (row-order accumulation, one
 * division).  p_list[k] >= 1.  orth_off/orth_q: for period p the ordered list of sub-periods
 * p/f (f prime, proper) to project out (Periods.py:209-214); ignored without PH_FLAG_ORTH. */
int ph_project_batch(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N,
                     const int32_t* p_list, int n_p,
                     const int32_t* orth_off, const int32_t* orth_q, int table_max_p,
                     unsigned flags, void* out);

/* ---- K2: fused all-p sweep (inner loops Periods.py:501-510 and :324-331) ----------------
 * out[w, p - p_lo] for p in [p_lo, p_hi] (inclusive), float64, shape (W, p_hi - p_lo + 1).
 * One launch per window batch; each window is read from HBM once and stays in LDS. */
int ph_sweep(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int p_lo, int p_hi,
             int mode, const int32_t* orth_off, const int32_t* orth_q, int table_max_p,
             unsigned flags, double* out);

/* ---- Periods._m_best_meta (Periods.py:456-601; m_best :408-430, m_best_gamma :432-454) --
 * periods (W, num) uint32, powers (W, num) float64, bases (W, num, N) dtype of x,
 * status (W) int32 (PH_ST_*).  Step 1 (the repeated all-p sweep, argmax, subtract) and
 * step 2 (factor refinement) both run on the device.  fac_off/fac_q: ordered proper
 * divisors (1 and p removed) of every p <= max_length, i.e. the iteration order of
 * get_factors(p, remove_1_and_n=True) at Periods.py:548-549.  Pass max_length < 0 for the
 * reference default floor(N/3). */
int ph_m_best(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int num,
              int min_length, int max_length, int gamma,
              const int32_t* orth_off, const int32_t* orth_q,
              const int32_t* fac_off, const int32_t* fac_q, int table_max_p,
              unsigned flags, uint32_t* periods, double* powers, void* bases, int32_t* status,
              int32_t* n_sweeps /* (W) all-p sweeps each window needed in step 1, or NULL */);

/* ---- Periods.small_to_large (Periods.py:246-287) ----------------------------------------
 * counts (W) int32 = number of accepted periods; periods (W, cap) int32; powers (W, cap)
 * float64; bases (W, cap, N) dtype of x or NULL.  Windows that accept more than `cap`
 * periods get status PH_ST_CAP, counts[w] holds the true count, and the call returns
 * PH_E_CAP so that the host can retry with a larger cap -- also with PH_FLAG_DEVICE: the
 * library then reads the batch's largest count back (one word; the call synchronises the
 * stream).  PH_FLAG_DEVICE | PH_FLAG_NOSYNC keeps the call asynchronous and returns PH_OK;
 * the caller must then inspect `status`.  n_periods < 0 = floor(N/2). */
int ph_small_to_large(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, double thresh,
                      int n_periods, const int32_t* orth_off, const int32_t* orth_q,
                      int table_max_p, unsigned flags, int cap, int32_t* counts,
                      int32_t* periods, double* powers, void* bases, int32_t* status);

/* ---- Periods.best_correlation (Periods.py:289-349) --------------------------------------
 * periods (W, num) uint32, norms (W, num) float64, bases (W, num, N).  max_length < 0 =
 * floor(N/3); candidate periods are 2 .. max_length-1 (exclusive bound, Periods.py:324). */
int ph_best_correlation(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int num,
                        int max_length, double ratio,
                        const int32_t* orth_off, const int32_t* orth_q, int table_max_p,
                        unsigned flags, uint32_t* periods, double* norms, void* bases,
                        int32_t* status);

/* ---- Periods.best_frequency (Periods.py:351-398) ----------------------------------------
 * num times: k = argmax |rfft(residual, win_size)| (first maximum, Periods.py:386-389),
 * p = round(2 win_size / k) (:390-391, round-half-even), project, store, subtract (:392-397).
 * The spectrum is an in-LDS radix-2 FFT when win_size is a power of two whose complex work array
 * fits the LDS (win_size <= 8192), Bluestein's chirp convolution on two such FFTs for any other
 * win_size up to about 5400, otherwise a direct real DFT over the first min(N, win_size) samples
 * (no FFT library; bins spread over the whole GPU); twiddles from float64 tables in every case; the projection honours PH_FLAG_TRUNC / PH_FLAG_ORTH (orth tables must cover
 * p <= 2 win_size).  win_size < 1 = N (:381-382).  Two launches per round; W <= 65535.
 * periods (W, num) uint32; powers (W, num) float64 = norm / ||data|| (:397-399); bases
 * (W, num, N).  status PH_ST_NO_PERIOD: the spectral peak was bin 0 (or the spectrum NaN) at
 * some iteration -- the reference divides by zero there and raises OverflowError; rows from that
 * iteration on are zero. */
int ph_best_frequency(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int win_size,
                      int num, const int32_t* orth_off, const int32_t* orth_q, int table_max_p,
                      unsigned flags, uint32_t* periods, double* powers, void* bases,
                      int32_t* status);

/* ---- RamanujanPeriods.find_periods (RamanujanPeriods.py:67-86 with :124-169) ------------
 * out (W, q_hi + 1) float64; entries below q_lo are zero (RamanujanPeriods.py:71).
 * Evaluated in float64 through the folded form (fold to S_q, Moebius-filter with the
 * integer Ramanujan sum c_q); the reference accumulates in float32, parity is 1e-5.
 * Any range: q_hi = N / 3 (the reference default, RamanujanPeriods.py:68-69) works for
 * N = 4096 .. 16384 and beyond, as long as one wavefront's strips (12 q_hi bytes) fit the LDS. */
int ph_ramanujan_norms(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int q_lo,
                       int q_hi, unsigned flags, double* out);

/* ---- RamanujanPeriods.project(x, basis) (RamanujanPeriods.py:124-131) ------------------
 * Arbitrary dictionary: x (N) float64, basis (rows, N) float64 -> out (rows, N) float32 with
 * out[i] = dot(x, basis[i]/max(basis[i])) * basis[i]/max(basis[i]). */
int ph_dict_project(ph_ctx* ctx, const double* x, const double* basis, int rows, int N,
                    unsigned flags, float* out);

/* ---- QOPeriods.find_periods, non-orthogonal / update_weights=True branch -----------------
 * (QOPeriods.py:373-596, get_subspaces :830-840, solve_quadratic :779-796) with the default
 * test function rms(reconstruction) > rms(data) * thresh.  The whole greedy loop runs on the
 * device, one workgroup per window: gamma sweep, phi-mass row bookkeeping, Gram matrix (closed-form
 * counts, never formed: regenerated inside the product) and right-hand side by folds, matrix-free preconditioned
 * conjugate-gradient solve of A A^T w = A x, reconstruction, residual.
 * Any N (the residual moves to an HBM workspace when it does not fit the LDS beside the solver).
 * periods/norms/keeps (W, num): dictionary blocks in the order found (period, gamma norm, rows
 * kept); counts (W, 2) = {periods the reference reports, blocks in the dictionary} (they differ
 * by one when the test function stopped the loop, QOPeriods.py:584-592); weights (W, kcap)
 * float64, rows of block b start at sum(keeps[:b]); residual (W, N) dtype of x.
 * kcap = capacity in dictionary rows per window (status PH_ST_CAP when exceeded). */
int ph_qo_find_periods(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int num,
                       double thresh, int min_length, int max_length, int kcap, unsigned flags,
                       uint32_t* periods, double* norms, int32_t* keeps, int32_t* counts,
                       double* weights, void* residual, int32_t* status);

/* *ok = 1 when ph_qo_find_periods can run windows of N samples of `dtype` with `kcap` dictionary
 * rows on this device (bookkeeping and the six work vectors of the conjugate-gradient solve fit the
 * workgroup's LDS; the window joins them there or moves to the HBM workspace), else 0 -- callers fall back to a host-driven loop instead of catching PH_E_ARG. */
int ph_qo_feasible(ph_ctx* ctx, int dtype, int N, int max_length, int kcap, int* ok);

/* ---- QOPeriods.get_best_period_orthogonal / eq_3 / auto_corr (QOPeriods.py:1122-1232) -----
 * powers (W, max_p) float64: the Muresan-Parks orthogonal period powers `pows` for q < max_p
 * (entry 0 is 0), divided by q when normalize != 0; autocorr (W, N) = auto_corr(x, k) for every
 * lag k, eq3 (W, max_p) = eq_3(x, q) -- both optional (NULL).  max_p < 0 = floor(N/2). */
int ph_orth_powers(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N, int max_p,
                   int normalize, unsigned flags, double* autocorr, double* eq3, double* powers);

/* ---- QOPeriods building blocks (QOPeriods.py:779-795) -----------------------------------
 * ph_fold_sums: W = A x for natural-basis rows -- out[w, off_k + j] = sum_{n = j (mod p_k)}
 * x[w, n], j < keep_k; row stride = sum(keep).  ph_tile_sum: reconstruction A^T w --
 * out[w, n] = sum_k wts[w, off_k + (n mod p_k)] (0 where n mod p_k >= keep_k). */
int ph_fold_sums(ph_ctx* ctx, const void* x, int dtype, int64_t W, int N,
                 const int32_t* p_list, const int32_t* keep, int n_p, unsigned flags,
                 double* out);
int ph_tile_sum(ph_ctx* ctx, const double* wts, int64_t W, int N, const int32_t* p_list,
                const int32_t* keep, int n_p, int dtype, unsigned flags, void* out);

#ifdef __cplusplus
}
#endif
#endif /* PERIODHIP_H */
