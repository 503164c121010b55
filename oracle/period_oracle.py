"""CPU oracle for the pyPeriod projection hot path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy *restatement* of the reference algorithm (woolgathering/pyPeriod @ v1).
It is the checker for the HIP path, never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it; nothing under ``pyperiod_amd/`` does, and the product fails loudly without its
    HIP library instead of falling back to this file;
  * it runs one 1-D window per call, like the reference.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against golden
vectors produced by importing the reference itself in the build container
(``tests/golden/make_golden.py``; shims of SURVEY.md section 8c) and against the known answers
of SURVEY.md section 4.  All ``file:line`` citations are into ``/root/reference/pyPeriod/``.

Third-party arithmetic underneath the reference is numpy (pad/reshape/add.reduce/mean/tile/
linalg.norm/linalg.solve); the reference pins numpy 1.19.2, this image has numpy 2.2 -- the
fixtures were generated with the latter.
"""

from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------------------
# integer helpers (Periods.py:33-84,121 ; RamanujanPeriods.py:14-39 ; QOPeriods.py:16-75)
# --------------------------------------------------------------------------------------


def primes_upto(limit: int = 10000) -> np.ndarray:
    """All primes <= limit (same result as get_primes, Periods.py:33-52)."""
    sieve = np.ones(limit + 1, dtype=bool)
    sieve[:2] = False
    for k in range(2, int(limit ** 0.5) + 1):
        if sieve[k]:
            sieve[k * k :: k] = False
    return np.flatnonzero(sieve)


PRIMES = set(int(v) for v in primes_upto(10000))  # Periods.py:121


def factor_set(n, remove_1_and_n: bool = False) -> set:
    """Divisor set of n with the *called* semantics get_factors(n, remove_1_and_n)
    (RamanujanPeriods.py:25-39; call sites Periods.py:209,548).

    The divisors are inserted in the order [1, n, 2, n/2, ...] exactly like the reference's
    ``reduce(list.__add__, ...)`` so that the CPython set iteration order -- which the
    reference's orthogonalisation (Periods.py:209) and m_best step 2 (Periods.py:549,570)
    depend on -- is inherited.
    """
    n = int(n)
    seq = []
    for i in range(1, int(n ** 0.5) + 1):
        if n % i == 0:
            seq.append(i)
            seq.append(n // i)
    out = set(seq)
    if remove_1_and_n:
        out.remove(1)
        out.remove(n)
    return out


def phi(n: int) -> int:
    """Euler totient (QOPeriods.py:16-43, RamanujanPeriods.py:14-22)."""
    return sum(1 for k in range(1, int(n) + 1) if math.gcd(int(n), k) == 1)


# --------------------------------------------------------------------------------------
# Periods.project / periodic_norm  (Periods.py:142-241)
# --------------------------------------------------------------------------------------


def fold_sums(data: np.ndarray, p: int, rows: int | None = None) -> np.ndarray:
    """S_p[j] = sum_{n = j (mod p)} data[n] accumulated row by row (Periods.py:171-176,194).

    ``np.add.reduce(axis=0)`` on the C-contiguous (R, p) rectangle adds row r into the p
    accumulators for r = 0..R-1 -- the same order as the reference's ``np.sum(cp, 0)``.
    """
    p = int(p)
    n = data.size
    r_full = -(-n // p)
    rect = np.pad(data, (0, r_full * p - n)).reshape(r_full, p)
    if rows is not None:
        rect = rect[:rows]
    return np.add.reduce(rect, axis=0)


def fold_counts(n: int, p: int) -> np.ndarray:
    """cnt_p[j] = #{m < n : m = j (mod p)} -- vectorised form of the `divs` loop
    (Periods.py:188-193)."""
    p = int(p)
    r_full = -(-n // p)
    short = r_full * p - n
    cnt = np.full(p, float(r_full))
    if short:
        cnt[p - short :] = r_full - 1
    return cnt


def project(
    data: np.ndarray,
    p: int = 2,
    trunc_to_integer_multiple: bool = False,
    orthogonalize: bool = False,
    return_single_period: bool = False,
) -> np.ndarray:
    """Periods.project (Periods.py:142-219)."""
    p = int(p)
    n = data.size
    r_full = -(-n // p)
    short = r_full * p - n
    if trunc_to_integer_multiple:
        # mean over the complete rows only (Periods.py:178-184)
        rect = np.pad(data, (0, short)).reshape(r_full, p)
        single = np.mean(rect if short == 0 else rect[:-1], 0)
    else:
        with np.errstate(invalid="ignore", divide="ignore"):
            single = fold_sums(data, p) / fold_counts(n, p)  # Periods.py:188-194
    projection = np.tile(single, n // p + 1)[:n]  # Periods.py:196-198
    if orthogonalize:
        # Periods.py:208-214: for every *prime* proper factor f (set order), remove the
        # projection of the running result onto p/f
        for f in factor_set(p, True):
            if f in PRIMES:
                projection = projection - project(
                    projection, p // f, trunc_to_integer_multiple, False
                )
    return projection[:p] if return_single_period else projection


def periodic_norm(x: np.ndarray, p=None) -> float:
    """Periods.periodic_norm (Periods.py:221-241)."""
    v = np.linalg.norm(x) / np.sqrt(len(x))
    if p:
        v = v / np.sqrt(p)
    return v


# --------------------------------------------------------------------------------------
# sweep helpers used by the parity tests (no single reference function; they restate the
# inner loops Periods.py:501-510 and Periods.py:324-331)
# --------------------------------------------------------------------------------------


def sweep_norms(data, p_lo, p_hi, gamma=False, trunc=False, orth=False) -> np.ndarray:
    """[periodic_norm(project(data,p)) for p in p_lo..p_hi]  (Periods.py:501-510)."""
    return np.array(
        [
            periodic_norm(project(data, p, trunc, orth), p if gamma else None)
            for p in range(p_lo, p_hi + 1)
        ]
    )


def sweep_maxabs(data, p_lo, p_hi) -> np.ndarray:
    """[max_s |S_p[s]| for p in p_lo..p_hi]  (Periods.py:324-331; builtin ``sum`` starts at
    int 0 and adds left to right, i.e. the same row-order accumulation as fold_sums)."""
    return np.array([np.max(np.abs(fold_sums(data, p))) for p in range(p_lo, p_hi + 1)])


# --------------------------------------------------------------------------------------
# Periods.small_to_large  (Periods.py:246-287)
# --------------------------------------------------------------------------------------


def small_to_large(data, thresh=0.1, n_periods=None, trunc=False, orth=False):
    periods, powers, bases = [], [], []
    data_norm = periodic_norm(data)  # :269
    residual = data.copy()
    if n_periods is None:
        n_periods = len(data) // 2  # :271-272
    for p in range(2, n_periods + 1):  # inclusive, :273
        base = project(residual, p, trunc, orth)
        trial = residual - base
        imposed = (periodic_norm(residual) - periodic_norm(trial)) / data_norm  # :278-280
        if imposed > thresh:  # strict, :281
            residual = trial
            periods.append(p)
            powers.append(imposed)
            bases.append(base)
    return periods, powers, bases


# --------------------------------------------------------------------------------------
# Periods.best_correlation  (Periods.py:289-349)
# --------------------------------------------------------------------------------------


def best_correlation(data, num=5, max_length=None, ratio=0.01, trunc=False, orth=False):
    n = len(data)
    if max_length is None:
        max_length = n // 3
    periods = np.zeros(num, dtype=np.uint32)
    norms = np.zeros(num)
    bases = np.zeros((num, n))
    og_norm = periodic_norm(data)
    old_norm = og_norm
    work = data.copy()
    for i in range(num):
        best_cor, best_p = 0, None
        for p in range(2, max_length):  # EXCLUSIVE upper bound, :324
            col = np.abs(fold_sums(work, p))
            s = int(np.argmax(col))  # first maximum == strict '>' scan over s, :327-331
            if col[s] > best_cor:
                best_cor, best_p = col[s], p
        base = project(work, best_p, trunc, orth)  # :334-339
        work = work - base  # unconditional, :340
        this_norm = periodic_norm(work)
        gain = (old_norm - this_norm) / og_norm
        if gain > ratio:  # :343
            periods[i], norms[i], bases[i] = best_p, gain, base
            old_norm = this_norm
    return periods, norms, bases


# --------------------------------------------------------------------------------------
# Periods.best_frequency  (Periods.py:351-398)
# --------------------------------------------------------------------------------------


def best_frequency(data, win_size=None, num=5, trunc=False, orth=False):
    n = len(data)
    if win_size is None:
        win_size = n
    periods = np.zeros(num, dtype=np.uint32)
    norms = np.zeros(num)
    bases = np.zeros((num, n))
    work = data.copy()
    for i in range(num):
        mags = np.abs(np.fft.rfft(work, win_size))
        p = int(np.round((2 * win_size) / np.argmax(mags)))  # :387-388
        base = project(work, p, trunc, orth)
        periods[i], norms[i], bases[i] = p, periodic_norm(base), base
        work = work - base
    return periods, norms / periodic_norm(data), bases


# --------------------------------------------------------------------------------------
# Periods._m_best_meta  (Periods.py:456-601)
# --------------------------------------------------------------------------------------


def m_best(data, num=5, max_length=None, min_length=2, gamma=False, trunc=False, orth=False, trace=None):
    """m_best (gamma=False, Periods.py:408-430) / m_best_gamma (gamma=True, :432-454).
    `trace` (a dict, test tooling only) receives the step-1 picks before step 2 reshuffles them."""
    n = len(data)
    if max_length is None:
        max_length = n // 3  # :485-486
    work = data.copy()
    periods = np.zeros(num, dtype=np.uint32)
    norms = np.zeros(num)
    bases = np.zeros((num, n))
    skip = set()

    # ---- step 1 (:494-537)
    i = 0
    repeats = 0
    gaps = []
    while i < num:
        top_norm, top_p, top_base = 0, 0, None
        second = 0.0
        for p in range(min_length, max_length + 1):
            base = project(work, p, trunc, orth)
            nrm = periodic_norm(base, p if gamma else None)
            if p not in skip:
                second = max(second, min(nrm, top_norm)) if nrm == nrm else second
            if nrm > top_norm and p not in skip:  # strict: lowest p wins ties, :512
                top_p, top_norm, top_base = p, nrm, base
        gaps.append((top_norm - second) / top_norm if top_norm > 0 else 0.0)  # test tooling (trace) only
        present = top_p in set(periods)
        if present and repeats < 10:  # :518-524
            idx = np.where(periods == top_p)[0]
            bases[idx] += top_base
            norms[idx] += top_norm
            repeats += 1
        elif present:  # :525-529
            skip.add(top_p)
            repeats = 0
        else:  # :530-535
            periods[i], norms[i], bases[i] = top_p, top_norm, top_base
            i += 1
            repeats = 0
        work = work - top_base  # always, :537

    if trace is not None:
        trace["step1_periods"], trace["step1_norms"] = periods.copy(), norms.copy()
        trace["step1_min_gap"] = min(gaps) if gaps else 1.0  # smallest relative lead of a winner over the runner-up

    # ---- step 2 (:540-598).  The `changed` flag is reset at the top of every inner
    # iteration (:544) and the inner loop can only end on an `i += 1` branch, so the outer
    # `while changed` runs exactly once.
    stale_p = max_length  # the loop variable `p` of :501 read again at :559,572
    base = None
    i = 0
    while i < num:
        top_norm, top_f, top_base = 0, None, None
        for f in factor_set(periods[i], True):  # set order, :548-549
            base = project(bases[i], f, trunc, orth)
            nrm = periodic_norm(base, stale_p if gamma else None)
            if nrm > top_norm:
                top_f, top_norm, top_base = f, nrm, base
        if top_f is not None and not np.any(periods == top_f):  # :565
            x_q = bases[i] - top_base
            n_big = top_norm
            # :569-572 -- the norm of the LAST factor's projection, not of x_q
            n_small = periodic_norm(base, stale_p if gamma else None)
            floor = min(norms)
            if (n_small + n_big) > (norms[num - 1] + norms[i]) and n_small > floor and n_big > floor:
                bases[i] = x_q
                norms[i] = n_small
                bases = np.insert(bases, i, top_base, 0)[:num]
                norms = np.insert(norms, i, n_big)[:num]
                periods = np.insert(periods, i, top_f)[:num]
                # i is NOT advanced (:581-594)
            else:
                i += 1
        else:
            i += 1

    return periods, norms / periodic_norm(data), bases  # :600-601


# --------------------------------------------------------------------------------------
# RamanujanPeriods  (RamanujanPeriods.py:67-86,124-169)
# --------------------------------------------------------------------------------------


def ramanujan_cq(q: int, s: int = 0, repetitions: int = 1) -> np.ndarray:
    """Real Ramanujan sum c_q(n), n < q (RamanujanPeriods.py:133-148): the coprime
    exponentials are accumulated one k at a time like the reference's inner loop."""
    q = int(q)
    n = np.arange(q)
    vec = np.zeros(q, dtype=complex)
    for k in range(1, q + 1):
        if math.gcd(k, q) == 1:
            vec = vec + np.exp(1j * 2 * np.pi * k * n / q)
    return np.real(np.tile(np.roll(vec, s), repetitions))


def ramanujan_cq_exact(q: int) -> np.ndarray:
    """Integer closed form c_q(n) = sum_{d | gcd(n,q)} mu(q/d) d  (Hoelder); used to check
    that the float form above is the integer table the HIP kernel uses."""
    q = int(q)

    def mobius(m):
        res, d = 1, 2
        while d * d <= m:
            if m % d == 0:
                m //= d
                if m % d == 0:
                    return 0
                res = -res
            d += 1
        return -res if m > 1 else res

    out = np.zeros(q, dtype=np.int64)
    for n in range(q):
        g = math.gcd(n, q)
        out[n] = sum(mobius(q // d) * d for d in range(1, g + 1) if g % d == 0)
    return out


def ramanujan_dictionary(q: int, n: int, normalize: bool = True) -> np.ndarray:
    """Cq_complete (RamanujanPeriods.py:156-169): q circular shifts tiled to n, L2-normalised."""
    cq = ramanujan_cq(q)
    reps = -(-n // q)
    mat = np.zeros((q, n))
    for i in range(q):
        mat[i] = np.tile(np.roll(cq, i), reps)[:n]
        if normalize:
            mat[i] /= np.linalg.norm(mat[i])
    return mat


def ramanujan_project(x: np.ndarray, basis: np.ndarray) -> np.ndarray:
    """RamanujanPeriods.project (:124-131): per-row correlate-and-scale, stored as float32."""
    out = np.zeros(basis.shape, dtype=np.float32)
    for i, row in enumerate(basis):
        row = row / np.max(row)
        out[i] = np.dot(x, row) * row
    return out


def ramanujan_find_periods(x, min_length=2, max_length=None) -> np.ndarray:
    """RamanujanPeriods.find_periods (:67-86) with select_periods=None."""
    if not max_length:
        max_length = len(x) // 3
    norms = np.zeros(max_length + 1)
    for q in range(min_length, max_length + 1):
        proj = ramanujan_project(x, ramanujan_dictionary(q, len(x)))
        out = np.sum(proj, 0)  # float32
        norms[q] = np.sum(np.power(out, 2))
    return norms


def ramanujan_norm_folded_q(x, q: int) -> float:
    """One entry of the folded form below: a = S_q (star) c_q / phi(q), o = a (*) c_q / phi(q),
    norms[q] = sum_j cnt_q[j] o_j^2."""
    q = int(q)
    c = ramanujan_cq_exact(q).astype(np.float64) / phi(q)
    s = fold_sums(x, q)
    idx = (np.arange(q)[None, :] - np.arange(q)[:, None]) % q  # idx[i, j] = (j - i) mod q
    a = (c[idx] * s[None, :]).sum(1)  # a_i = sum_j S[j] c((j-i) mod q)
    o = (c[idx] * a[:, None]).sum(0)  # o_j = sum_i a_i c((j-i) mod q)
    return float(np.sum(fold_counts(len(x), q) * o * o))


def ramanujan_norms_folded(x, min_length=2, max_length=None) -> np.ndarray:
    """fp64 folded form of the same quantity.  This is what the HIP kernel
    evaluates; it differs from the float32 reference path by ~1e-7 relative."""
    n = len(x)
    if not max_length:
        max_length = n // 3
    norms = np.zeros(max_length + 1)
    for q in range(min_length, max_length + 1):
        norms[q] = ramanujan_norm_folded_q(x, q)
    return norms


def ramanujan_find_periods_with_weights(x, min_length=2, max_length=None, thresh=0.2, norms=None):
    """RamanujanPeriods.find_periods_with_weights (:88-122) as intended: the v1 tree dies on a
    missing ``_k`` and unpacks solve_quadratic's (weights, reconstruction) the wrong way round
    (:109); tests/golden/make_golden.py (shim 4) runs the reference with those two repaired.
    ``norms`` may be handed in (the float32-accumulated reference values decide the threshold
    test at :95-99; by default the fp64 folded form is used)."""
    if norms is None:
        norms = ramanujan_norms_folded(x, min_length, max_length)
    periods = np.argwhere(norms / np.abs(np.max(norms)) > thresh).flatten()  # :95-99
    a, dims = qo_get_subspaces(periods, len(x))  # :106-108
    w, recon = qo_solve_quadratic(x, a)  # :109-111
    out = {"periods": periods, "norms": norms[periods], "subspaces": a, "weights": w, "basis_dictionary": dims}
    return out, x - recon


# --------------------------------------------------------------------------------------
# QOPeriods pieces on the path (QOPeriods.py:598-643,743-852,940-1003)
# --------------------------------------------------------------------------------------


def qo_subspace_dims(periods, n: int) -> dict:
    """Row bookkeeping of get_subspaces (QOPeriods.py:830-840): period q keeps as many
    natural-basis rows as the Euler-phi mass its divisors add to the running divisor set."""
    dims = {}
    seen = set()
    prev = 0
    for q in periods:
        seen = seen.union(factor_set(int(q)))
        total = int(np.sum([phi(r) for r in seen]))
        dims[str(q)] = total - prev
        prev = total
    return dims


def qo_natural_rows(p: int, n: int, keep=None) -> np.ndarray:
    """Pp(p, N, keep, 'natural') (QOPeriods.py:940-1003): row i is the indicator of
    n = i (mod p); only the first `keep` rows are retained."""
    p = int(p)
    mat = (np.arange(n)[None, :] % p == np.arange(p)[:, None]).astype(np.float64)
    return mat[:keep] if keep else mat


def qo_get_subspaces(periods, n: int):
    dims = qo_subspace_dims(periods, n)
    blocks = [qo_natural_rows(int(q), n, keep) for q, keep in dims.items()]
    a = np.vstack(blocks) if blocks else np.zeros((0, n))
    return a, dims


def qo_solve_quadratic(x: np.ndarray, a: np.ndarray):
    """solve_quadratic(type='solve', window=None) (QOPeriods.py:779-796)."""
    gram = a @ a.T
    rhs = a @ x
    w = np.linalg.solve(gram, rhs)
    return w, a.T @ w


def qo_find_periods(data, num, thresh, min_length=2, max_length=None):
    """QOPeriods.find_periods, non-orthogonal / update_weights=True branch
    (QOPeriods.py:373-596 with :468-478 and :510-522) -- the only branch that runs in the v1
    tree."""
    n = len(data)
    if max_length is None:
        max_length = n // 3
    periods = np.zeros(num, dtype=np.uint32)
    norms = np.zeros(num)
    res = data.copy()
    rms = lambda v: np.sqrt(np.sum(np.power(v, 2)) / len(v))
    out = {"periods": [], "norms": [], "subspaces": [], "weights": [], "basis_dictionary": {}}
    if np.sum(np.abs(data)) <= 1e-16:  # :394-406
        return (
            {
                "periods": np.array([1]),
                "norms": np.array([0]),
                "subspaces": np.ones((1, n)),
                "weights": np.array([0]),
                "basis_dictionary": {"1": n},
            },
            np.zeros(n),
        )
    recon = None
    nonzero = periods[:0]
    for i in range(num):
        if i == 0 or rms(recon) > rms(data) * thresh:  # default test_function, :391,418
            best_p, best_norm = 0, 0
            for p in range(min_length, max_length + 1):  # :470-478
                nrm = periodic_norm(project(res, p, False, False), p)
                if nrm > best_norm:
                    best_p, best_norm = p, nrm
            periods[i], norms[i] = best_p, best_norm
            nonzero = periods[periods > 0]
            try:
                a, dims = qo_get_subspaces(nonzero, n)
                w, recon = qo_solve_quadratic(data, a)
            except np.linalg.LinAlgError:  # :552-559
                break
            res = data - recon
            out = {
                "periods": nonzero,
                "norms": norms[: len(nonzero)],
                "subspaces": a,
                "weights": w,
                "basis_dictionary": dims,
            }
        else:  # :560-594
            a, dims = qo_get_subspaces(nonzero, n)
            w, recon = qo_solve_quadratic(data, a)
            out = {
                "periods": nonzero[:-1],
                "norms": norms[: len(nonzero) - 1],
                "subspaces": a,
                "weights": w,
                "basis_dictionary": dims,
            }
            break
    return out, res


# --------------------------------------------------------------------------------------
# Orthogonal period powers  (QOPeriods.py:1122-1232)
# --------------------------------------------------------------------------------------


def auto_corr(x, k: int) -> float:
    """QOPeriods.auto_corr (:1151-1173)."""
    n = len(x)
    return np.sum(x[0 : n - k] * x[k:n])


def eq_3(x, p: int) -> float:
    """QOPeriods.eq_3 (:1122-1149)."""
    n = len(x)
    second = 0
    for l in range(1, n // p):
        second += auto_corr(x, int(l * p))
    return (p / n) * (auto_corr(x, 0) + 2 * second)


def orth_powers(x, max_p=None, normalize=False) -> np.ndarray:
    """get_best_period_orthogonal(..., return_powers=True) (:1175-1225)."""
    if max_p is None:
        max_p = len(x) // 2
    q_all = np.arange(1, max_p)
    pows = np.zeros(q_all[-1] + 1)
    for q in q_all:
        pows[q] = max(eq_3(x, int(q)), 0)
        for f in factor_set(int(q)):
            if f != q:
                pows[q] -= pows[f]
    pows[pows < 0] = 0
    if normalize:
        pows[1:] = pows[1:] / q_all
    return pows


def best_period_orthogonal(x, max_p=None, normalize=False) -> int:
    """get_best_period_orthogonal(..., return_powers=False) (:1226-1232)."""
    pows = orth_powers(x, max_p, normalize)
    k = int(np.argmax(pows))
    return k if k > 0 else 1

