"""Seeded randomised differential tests: HIP path vs the CPU oracle over random window lengths,
period ranges, flags and algorithm parameters (ragged sizes exercise every tail path of the
chunked / multi-period folds)."""

import warnings

import numpy as np
import pytest

from conftest import rel_err
from oracle import period_oracle as po
from pyperiod_amd.synth import multi_sinusoid_batch

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as ge

    ge.build()
    from pyperiod_amd import default_engine

    return default_engine()


@pytest.fixture(autouse=True)
def _quiet():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def _windows(rng, w, n):
    kind = rng.integers(0, 3)
    if kind == 0:
        return multi_sinusoid_batch(int(rng.integers(0, 10_000)), w, n) if n >= 88 else rng.standard_normal((w, n))
    if kind == 1:
        return rng.standard_normal((w, n))
    t = np.arange(n)
    return np.stack([np.sin(2 * np.pi * t / rng.integers(3, max(4, n // 4))) + 0.1 * rng.standard_normal(n) for _ in range(w)])


def test_random_sweeps(eng):
    from pyperiod_amd import _ffi

    rng = np.random.default_rng(2024)
    for trial in range(40):
        n = int(rng.integers(5, 3000))
        w = int(rng.integers(1, 4))
        x = _windows(rng, w, n)
        p_lo = int(rng.integers(1, max(2, min(n, 200))))
        p_hi = int(rng.integers(p_lo, max(p_lo + 1, min(n + 5, 1500))))
        mode = int(rng.integers(0, 3))
        got = eng.sweep(x, p_lo, p_hi, mode)
        for i in range(w):
            if mode == 2:
                want = po.sweep_maxabs(x[i], p_lo, p_hi)
                lo = 1 if p_lo == 1 else 0
                assert np.array_equal(got[i, lo:], want[lo:]), (trial, n, p_lo, p_hi)
            else:
                want = po.sweep_norms(x[i], p_lo, p_hi, gamma=(mode == 1))
                assert rel_err(got[i], want) < TOL, (trial, n, p_lo, p_hi, mode)


def test_random_flagged_sweeps_and_projections(eng):
    from pyperiod_amd import _ffi

    rng = np.random.default_rng(77)
    for trial in range(12):
        n = int(rng.integers(20, 700))
        x = _windows(rng, 2, n)
        trunc, orth = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        p_hi = int(rng.integers(2, n // 2 + 2))
        got = eng.sweep(x, 2, p_hi, _ffi.PH_SWEEP_NORM, trunc, orth)
        plist = sorted(set(int(v) for v in rng.integers(2, n + 3, size=6)))
        proj = eng.project_batch(x, plist, trunc, orth)
        for i in range(2):
            assert rel_err(got[i], po.sweep_norms(x[i], 2, p_hi, trunc=trunc, orth=orth)) < TOL, (trial, n, trunc, orth)
            for k, p in enumerate(plist):
                assert np.array_equal(proj[i, k], po.project(x[i], p, trunc, orth), equal_nan=True), (trial, n, p, trunc, orth)


def test_random_m_best(eng):
    rng = np.random.default_rng(5)
    for trial in range(14):
        n = int(rng.integers(60, 1400))
        w = int(rng.integers(1, 4))
        x = _windows(rng, w, n)
        num = int(rng.integers(1, 7))
        min_len = int(rng.integers(2, 6))
        max_len = int(rng.integers(min_len + 3, max(min_len + 4, n // 2)))
        gamma = bool(rng.integers(0, 2))
        trunc, orth = (bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) if trial % 3 == 0 else (False, False)
        per, pw, bs, st = eng.m_best(x, num, max_len, min_len, gamma, trunc, orth)
        for i in range(w):
            try:
                rper, rpw, rbs = po.m_best(x[i], num, max_len, min_len, gamma, trunc, orth)
            except TypeError:
                assert st[i] != 0
                continue
            assert st[i] == 0, (trial, i)
            assert np.array_equal(per[i], rper), (trial, n, num, min_len, max_len, gamma, trunc, orth)
            assert rel_err(pw[i], rpw) < TOL and rel_err(bs[i], rbs) < TOL


def test_random_small_to_large_and_best_correlation(eng):
    rng = np.random.default_rng(11)
    for trial in range(12):
        n = int(rng.integers(40, 1500))
        w = int(rng.integers(1, 4))
        x = _windows(rng, w, n)
        thresh = float(rng.choice([0.005, 0.02, 0.05, 0.1, 0.3]))
        n_per = None if trial % 2 else int(rng.integers(2, n // 2 + 1))
        trunc, orth = (bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) if trial % 4 == 0 else (False, False)
        counts, per, pw, bs, st = eng.small_to_large(x, thresh, n_per, trunc, orth, cap=4)
        for i in range(w):
            rper, rpw, rbs = po.small_to_large(x[i], thresh, n_per, trunc, orth)
            k = counts[i]
            assert list(per[i, :k]) == rper, (trial, n, thresh, n_per, trunc, orth)
            assert rel_err(pw[i, :k], rpw) < TOL
            if k:
                assert rel_err(bs[i, :k], np.array(rbs)) < TOL
        num = int(rng.integers(1, 4))
        max_len = int(rng.integers(4, max(5, n // 3 + 2)))
        ratio = float(rng.choice([0.0, 0.01, 0.05]))
        per, nr, bs, st = eng.best_correlation(x, num, max_len, ratio)
        for i in range(w):
            rper, rnr, rbs = po.best_correlation(x[i], num, max_len, ratio)
            assert np.array_equal(per[i], rper), (trial, n, num, max_len, ratio)
            assert rel_err(nr[i], rnr) < TOL and rel_err(bs[i], rbs) < TOL


def test_random_ramanujan_and_qo(eng):
    rng = np.random.default_rng(3)
    for trial in range(8):
        n = int(rng.integers(30, 2500))
        x = _windows(rng, 2, n)
        q_hi = int(rng.integers(2, min(n // 2, 400) + 1))
        q_lo = int(rng.integers(1, q_hi + 1))
        got = eng.ramanujan_norms(x, q_lo, q_hi)
        for i in range(2):
            want = po.ramanujan_norms_folded(x[i], q_lo, q_hi)
            scale = max(np.max(np.abs(want)), 1e-300)
            assert np.max(np.abs(got[i] - want)) / scale < 1e-9, (trial, n, q_lo, q_hi)
    for trial in range(6):
        n = int(rng.integers(200, 1200))
        x = multi_sinusoid_batch(int(rng.integers(0, 1000)), 2, n)
        num = int(rng.integers(1, 5))
        thresh = float(rng.choice([0.05, 0.2, 0.5]))
        lo = int(rng.integers(2, 8))
        hi = int(rng.integers(lo + 5, n // 3 + 1))
        per, nrm, keeps, counts, wts, resid, st = eng.qo_find_periods(x, num, thresh, lo, hi)
        for i in range(2):
            out, res = po.qo_find_periods(x[i], num, thresh, lo, hi)
            nrep, nb = counts[i]
            assert st[i] == 0
            assert np.array_equal(per[i, :nrep], np.asarray(out["periods"])), (trial, n, num, thresh, lo, hi)
            assert list(keeps[i, :nb]) == list(out["basis_dictionary"].values())
            k = int(keeps[i, :nb].sum())
            assert rel_err(wts[i, :k], out["weights"]) < 1e-7 and rel_err(resid[i], res) < 1e-7
