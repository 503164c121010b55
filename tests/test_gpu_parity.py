"""GPU parity tests: the HIP path (through the C ABI / ctypes engine) against the CPU oracle
on the same seeded inputs, against the committed golden vectors from the reference, and --
at BASELINE.json's full sizes -- through size-independent properties.

Bars (BASELINE.json): integer period lists bit-exact; fp64 powers / bases / norms within
1e-10 relative; Periods.project (non-orth) bit-identical; Ramanujan norms within 1e-5
relative (the reference accumulates in float32, RamanujanPeriods.py:127).
"""

import warnings

import numpy as np
import pytest

from conftest import elem_err, rel_err
from oracle import period_oracle as po
from pyperiod_amd.synth import multi_sinusoid_batch, multi_sinusoid_window, readme_window

pytestmark = pytest.mark.gpu

TOL = 1e-10
FLAGS = [(False, False), (True, False), (False, True), (True, True)]


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


# ------------------------------------------------------------------------------ K1 project
def test_project_matches_reference_golden(eng, golden):
    g = golden("project")
    for n in (10, 97, 240, 4096):
        x = g[f"x_{n}"]
        plist = [p for p in (2, 3, 7, 12, 64, 97, n // 2) if 2 <= p <= n]
        for trunc, orth in FLAGS:
            out = eng.project_batch(x[None, :], plist, trunc, orth)[0]
            for k, p in enumerate(plist):
                key = f"n{n}_p{p}_t{int(trunc)}_o{int(orth)}"
                want = g[key] if n <= 240 else np.tile(g[key + "_single"], n // p + 1)[:n]
                assert np.array_equal(out[k], want), key  # bit-identical, all four modes


def test_project_batch_bit_exact_vs_oracle(eng):
    n = 1000
    x = multi_sinusoid_batch(20, 5, n)
    plist = list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 333, 499, 500, 501, 999, 1000]
    for trunc, orth in FLAGS:
        out = eng.project_batch(x, plist, trunc, orth)
        for w in range(x.shape[0]):
            for k, p in enumerate(plist):
                if p == 1 and orth:
                    continue  # the reference raises KeyError here (set.remove(1) twice)
                want = po.project(x[w], p, trunc, orth)
                if p == 1 and not orth:
                    # a (N, 1) rectangle reduces along its contiguous axis, where numpy sums
                    # pairwise; p = 1 (the mean) is never part of a sweep
                    assert rel_err(out[w, k], want) < 1e-12
                else:
                    assert np.array_equal(out[w, k], want, equal_nan=True), (trunc, orth, w, p)
    single = eng.project_batch(x, plist, False, False, single=True)
    for k, p in enumerate(plist):
        assert np.array_equal(single[:, k, :p], eng.project_batch(x, [p])[:, 0, :p])
        assert not single[:, k, p:].any()


def test_project_period_longer_than_window(eng):
    x = multi_sinusoid_window(3, 50)
    out = eng.project_batch(x[None, :], [64, 51, 50, 49])[0]
    for k, p in enumerate([64, 51, 50, 49]):
        assert np.array_equal(out[k], po.project(x, p), equal_nan=True)
    t = eng.project_batch(x[None, :], [64], True, False)[0, 0]
    assert np.isnan(t).all()  # np.mean over zero complete rows (Periods.py:182-184)


def test_project_float32_window(eng):
    x = multi_sinusoid_batch(0, 3, 512, dtype=np.float32)
    out = eng.project_batch(x, [7, 64, 100])
    assert out.dtype == np.float32
    for w in range(3):
        for k, p in enumerate([7, 64, 100]):
            assert rel_err(out[w, k], po.project(x[w].astype(np.float64), p)) < 1e-5


def test_periodic_norm(eng):
    x = multi_sinusoid_batch(5, 4, 777)
    got = eng.periodic_norm(x)
    for w in range(4):
        assert abs(got[w] - po.periodic_norm(x[w])) <= 1e-14 * got[w]
    got = eng.periodic_norm(x, 9)
    for w in range(4):
        assert abs(got[w] - po.periodic_norm(x[w], 9)) <= 1e-14 * got[w]
    assert abs(eng.periodic_norm(np.arange(10.0)[None, :])[0] - 5.338539126015656) < 1e-14


# ------------------------------------------------------------------------------ K2 sweep
def test_sweep_matches_reference_golden(eng, golden):
    from pyperiod_amd import _ffi

    g = golden("sweep")
    n = 4096
    x = multi_sinusoid_batch(0, 4, n)
    plain = eng.sweep(x, 2, n // 3, _ffi.PH_SWEEP_NORM)
    gamma = eng.sweep(x, 2, n // 3, _ffi.PH_SWEEP_NORM_GAMMA)
    for w in range(4):
        assert rel_err(plain[w], g[f"plain_w{w}"]) < TOL
        assert rel_err(gamma[w], g[f"gamma_w{w}"]) < TOL
        # entry by entry (north_star: 1e-10 *relative*): weak small-p entries meet the same bar
        assert elem_err(plain[w], g[f"plain_w{w}"]) < TOL and elem_err(gamma[w], g[f"gamma_w{w}"]) < TOL
    maxabs = eng.sweep(x[:1], 2, n // 3, _ffi.PH_SWEEP_MAXABS)[0]
    assert np.array_equal(maxabs, g["maxabs_w0"])  # row-order sums: bit-identical
    for trunc, orth in FLAGS[1:]:
        got = eng.sweep(x[:1], 2, n // 3, _ffi.PH_SWEEP_NORM, trunc, orth)[0]
        assert rel_err(got, g[f"plain_w0_t{int(trunc)}_o{int(orth)}"]) < TOL


def test_sweep_ragged_sizes_vs_oracle(eng):
    from pyperiod_amd import _ffi

    for n, p_lo, p_hi in ((97, 1, 97), (1000, 2, 500), (130, 60, 140)):
        x = multi_sinusoid_batch(9, 3, n)
        got = eng.sweep(x, p_lo, p_hi, _ffi.PH_SWEEP_NORM)
        gam = eng.sweep(x, p_lo, p_hi, _ffi.PH_SWEEP_NORM_GAMMA)
        mab = eng.sweep(x, p_lo, p_hi, _ffi.PH_SWEEP_MAXABS)
        for w in range(3):
            assert rel_err(got[w], po.sweep_norms(x[w], p_lo, p_hi)) < TOL
            assert rel_err(gam[w], po.sweep_norms(x[w], p_lo, p_hi, gamma=True)) < TOL
            want = po.sweep_maxabs(x[w], p_lo, p_hi)
            lo = 1 if p_lo == 1 else 0  # p = 1: numpy sums the contiguous axis pairwise
            assert np.array_equal(mab[w, lo:], want[lo:])
            assert rel_err(mab[w, :lo], want[:lo]) < 1e-13


# ------------------------------------------------------------------------------ m_best
@pytest.mark.parametrize("name,gamma", [("m_best", False), ("m_best_gamma", True)])
def test_m_best_matches_reference_golden(eng, golden, name, gamma):
    g = golden("m_best")
    x = multi_sinusoid_batch(0, 4, 4096)
    per, pw, bs, st = eng.m_best(x, 10, None, 2, gamma)
    assert per.dtype == np.uint32 and not st.any()
    for w in range(4):
        assert np.array_equal(per[w], g[f"{name}_w{w}_periods"]), w
        assert rel_err(pw[w], g[f"{name}_w{w}_powers"]) < TOL and elem_err(pw[w], g[f"{name}_w{w}_powers"]) < TOL
    assert rel_err(bs[1], g[f"{name}_w1_bases"]) < TOL
    x = multi_sinusoid_batch(4, 2, 1500)
    per, pw, bs, st = eng.m_best(x, 6, 300, 3, gamma)
    for k, w in enumerate((4, 5)):
        assert np.array_equal(per[k], g[f"{name}_n1500_w{w}_periods"])
        assert rel_err(pw[k], g[f"{name}_n1500_w{w}_powers"]) < TOL
        assert rel_err(bs[k], g[f"{name}_n1500_w{w}_bases"]) < TOL
    per, pw, bs, st = eng.m_best(readme_window(2000, 0)[None, :], 10, None, 2, gamma)
    assert np.array_equal(per[0], g[f"{name}_c1_periods"])
    assert rel_err(pw[0], g[f"{name}_c1_powers"]) < TOL and rel_err(bs[0], g[f"{name}_c1_bases"]) < TOL
    for trunc, orth in FLAGS[1:]:
        tag = f"{name}_n900_t{int(trunc)}_o{int(orth)}"
        per, pw, bs, st = eng.m_best(multi_sinusoid_window(6, 900)[None, :], 5, None, 2, gamma, trunc, orth)
        assert np.array_equal(per[0], g[tag + "_periods"]), tag
        assert rel_err(pw[0], g[tag + "_powers"]) < TOL and rel_err(bs[0], g[tag + "_bases"]) < TOL


@pytest.mark.parametrize("name,gamma", [("m_best_gamma", True), ("m_best", False)])
def test_m_best_step2_splits_match_reference_golden(eng, golden, name, gamma):
    """Fixtures in which step 2 really splits a row (np.insert fires in the reference,
    Periods.py:581-594): in gamma mode they pin the stale-`p` divisor of Periods.py:559,572."""
    g = golden("m_best_split")
    for n, ml, num, w in g["cases"]:
        tag = f"{name}_n{n}_ml{ml}_num{num}_w{w}"
        per, pw, bs, st = eng.m_best(multi_sinusoid_window(int(w), int(n))[None, :], int(num), int(ml), 2, gamma)
        assert st[0] == 0 and np.array_equal(per[0], g[tag + "_periods"]), tag
        assert rel_err(pw[0], g[tag + "_powers"]) < TOL and elem_err(pw[0], g[tag + "_powers"]) < 1e-9, tag
        assert rel_err(bs[0], g[tag + "_bases"]) < TOL, tag


def test_m_best_batch_vs_oracle(eng):
    x = multi_sinusoid_batch(40, 6, 768)
    for gamma in (False, True):
        per, pw, bs, st = eng.m_best(x, 5, None, 2, gamma)
        for w in range(6):
            rper, rpw, rbs = po.m_best(x[w], 5, gamma=gamma)
            assert np.array_equal(per[w], rper), (gamma, w)
            assert rel_err(pw[w], rpw) < TOL and rel_err(bs[w], rbs) < TOL


def test_m_best_exhausted_range_boundary(eng):
    """DESIGN.md section 3, known limit: max_length = 7 offers six candidate periods; once 6 and 4 are removed
    the projections onto 2 and 3 are zero up to rounding, so a FIFTH step-1 pick is BLAS-nrm2 rounding noise in
    the reference (norm ~1e-17) and not reproducible.  Everything up to that pick is: asking for four periods
    agrees bit for bit (including the step-2 split the noise pick can block or allow when five are asked for)."""
    n, t = 672, np.arange(672)
    for seed in (2, 3, 4):
        rng = np.random.default_rng(seed)
        x = np.sin(2 * np.pi * t / int(rng.integers(9, 160))) + 0.05 * rng.standard_normal(n)
        tr = {}
        po.m_best(x, 5, 7, 2, False, trace=tr)
        assert tr["step1_norms"].min() < 1e-12 * tr["step1_norms"].max()  # the case is what the docstring says
        rper, rpw, rbs = po.m_best(x, 4, 7, 2, False, trace=tr)
        assert tr["step1_norms"].min() > 1e-6 * tr["step1_norms"].max()
        per, pw, bs, st = eng.m_best(x[None, :], 4, 7, 2, False)
        assert st[0] == 0 and np.array_equal(per[0], rper), seed
        assert rel_err(pw[0], rpw) < TOL and rel_err(bs[0], rbs) < TOL, seed


def test_m_best_reference_runs_out_of_candidates_on_an_exact_zero(eng):
    """The mirror image of the limit above (found by tools/fuzz_gpu.py, seed 9107): 18 quarter-integer samples, num = 4,
    max_length = 5.  Periods 3, 4 and 5 are picked, repeated and blacklisted; the projection onto 2 (contained in the one
    onto 4) is rounding noise by then -- 1e-17 in most sweeps, EXACTLY zero in the reference's order of additions at the
    sweep that matters, so the reference finds no candidate and raises (Periods.py:520).  The device adds in another
    order: it may report the same failure or make a fourth pick at rounding-noise level; the three real picks must be
    the reference's either way (asking for three periods agrees bit for bit)."""
    x = np.array([1.25, -0.75, -0.5, -0.75, -0.5, 0.75, -2.0, 1.0, 0.75, -0.25, -0.25, -0.25, 1.0, 0.0, -0.75, 1.0, 0.75, -0.5])
    with pytest.raises(Exception):
        po.m_best(x, 4, 5, 2, False)
    rper, rpw, rbs = po.m_best(x, 3, 5, 2, False)
    per3, pw3, bs3, st3 = eng.m_best(x[None, :], 3, 5, 2, False)
    assert st3[0] == 0 and np.array_equal(per3[0], rper)
    assert rel_err(pw3[0], rpw) < TOL and rel_err(bs3[0], rbs) < TOL
    per, pw, bs, st = eng.m_best(x[None, :], 4, 5, 2, False)
    if st[0] == 0:
        assert np.min(np.abs(pw[0])) < 1e-10 * np.max(np.abs(pw[0]))  # the fourth pick is noise, and its power says so


def test_m_best_periods_beyond_two_thirds_of_the_window(eng, golden):
    """max_length well above the default N/3 (reference fixture `m_best_large_p`): rows whose period exceeds
    2N/3 take step 2's tiled-row path (the compact p-vector plus its zero row no longer fits the row buffer;
    the first case really splits 454 -> 227 there), periods above N/2 have single-sample residues.
    (max_length = N - 1 itself is not a usable case: once p = N - 1 is removed the residual is two samples and
    every later period ties exactly.)"""
    g = golden("m_best_large_p")
    for n, ml, num, _ in g["cases"]:
        x = g[f"x_n{n}"]
        for name, gamma in (("m_best", False), ("m_best_gamma", True)):
            tag = f"{name}_n{n}_ml{ml}_num{num}"
            per, pw, bs, st = eng.m_best(x[None, :], int(num), int(ml), 2, gamma)
            assert st[0] == 0 and np.array_equal(per[0], g[tag + "_periods"]), (tag, per[0])
            assert rel_err(pw[0], g[tag + "_powers"]) < TOL and rel_err(bs[0], g[tag + "_bases"]) < TOL, tag


def test_m_best_zero_window_reports_status(eng):
    from pyperiod_amd import Periods, _ffi

    x = np.zeros((2, 256))
    x[1] = multi_sinusoid_window(0, 256)
    per, pw, bs, st = eng.m_best(x, 3)
    assert st[0] == _ffi.PH_ST_NO_PERIOD and st[1] == 0
    with pytest.raises(TypeError):  # the reference raises UFuncTypeError (a TypeError) at Periods.py:520
        Periods().m_best(np.zeros(256), num=3)


# ------------------------------------------------------------------------------ small_to_large
def test_small_to_large_matches_reference_golden(eng, golden):
    from pyperiod_amd import Periods

    g = golden("small_to_large")
    per, pw, bs = Periods().small_to_large(readme_window(2000, 0), thresh=0.1)  # BASELINE config 1
    assert per == list(g["c1_periods"]) and all(isinstance(v, int) for v in per)
    assert rel_err(pw, g["c1_powers"]) < TOL and rel_err(np.array(bs), g["c1_bases"]) < TOL
    x = multi_sinusoid_batch(0, 4, 4096)
    counts, per, pw, bs, st = eng.small_to_large(x, 0.05)  # config 4 unit
    for w in range(4):
        k = counts[w]
        assert list(per[w, :k]) == list(g[f"w{w}_periods"]), w
        assert rel_err(pw[w, :k], g[f"w{w}_powers"]) < TOL
    assert rel_err(bs[1, : counts[1]], g["w1_bases"]) < TOL
    for trunc, orth in FLAGS[1:]:
        tag = f"n1200_t{int(trunc)}_o{int(orth)}"
        per, pw, bs = Periods(trunc, orth).small_to_large(multi_sinusoid_window(2, 1200), thresh=0.05)
        assert per == list(g[tag + "_periods"]), tag
        assert rel_err(pw, g[tag + "_powers"]) < TOL
        assert rel_err(np.array(bs).reshape(len(per), 1200), g[tag + "_bases"]) < TOL
    per, pw, _ = Periods().small_to_large(multi_sinusoid_window(3, 600), thresh=0.02, n_periods=100)
    assert per == list(g["n600_np100_periods"]) and rel_err(pw, g["n600_np100_powers"]) < TOL


def test_small_to_large_device_pointer_cap_overflow_is_reported(eng):
    """PH_FLAG_DEVICE callers: a too small `cap` must not pass silently."""
    import torch

    from pyperiod_amd import _ffi

    x = multi_sinusoid_batch(60, 3, 500)
    xd = torch.from_numpy(x).cuda()
    counts, per, pw, bs, st = eng.small_to_large(xd, 0.001, cap=2)  # the engine sees PH_E_CAP and retries
    assert per.shape[1] >= int(counts.max()) > 2 and int(st.abs().max()) == 0
    for w in range(3):
        rper, rpw, _ = po.small_to_large(x[w], 0.001)
        assert list(per[w, : int(counts[w])].cpu().numpy()) == rper
    counts, per, pw, bs, st = eng.small_to_large(xd, 0.001, cap=2, nosync=True)  # opt-out: status says so
    assert per.shape[1] == 2 and (st.cpu().numpy() == _ffi.PH_ST_CAP).all()
    with pytest.raises(_ffi.CapacityError):  # the raw ABI returns PH_E_CAP for device pointers too
        c2 = torch.empty(3, dtype=torch.int32, device="cuda")
        p2 = torch.empty((3, 2), dtype=torch.int32, device="cuda")
        w2 = torch.empty((3, 2), dtype=torch.float64, device="cuda")
        s2 = torch.empty(3, dtype=torch.int32, device="cuda")
        _ffi.check(eng._lib.ph_small_to_large(eng._ctx, xd.data_ptr(), _ffi.PH_F64, 3, 500, 0.001, -1, None, None, 0,
                                             _ffi.PH_FLAG_DEVICE, 2, c2.data_ptr(), p2.data_ptr(), w2.data_ptr(), None, s2.data_ptr()))


def test_plan_info_and_feasibility_helpers(eng):
    n_pass, n_per = eng.sweep_plan_info(2, 1365)
    assert n_per == 1364 and 600 < n_pass < 1364  # multi-period passes: fewer passes than periods
    assert eng.sweep_plan_info(2, 63) == (32, 62)  # below 64: chains L, L/2, L/4, ... share one row-split pass (round 4)
    assert eng.qo_feasible(4096, np.float64, 512) and eng.qo_feasible(40000, np.float64, 512)  # long windows: HBM residual
    assert not eng.qo_feasible(4096, np.float64, 4096)  # beyond the row capacity of the workspace


def test_empty_batch_returns_empty_outputs(eng):
    """W == 0 (a trailing rank's shard, pyperiod_amd/dist.py): no launch, correctly shaped outputs."""
    x = np.zeros((0, 256))
    per, pw, bs, st = eng.m_best(x, 3)
    assert per.shape == (0, 3) and pw.shape == (0, 3) and bs.shape == (0, 3, 256) and st.shape == (0,)
    counts, sper, spw, sbs, sst = eng.small_to_large(x, 0.05, cap=8)
    assert counts.shape == (0,) and sper.shape == (0, 8)
    assert eng.sweep(x, 2, 50).shape == (0, 49) and eng.ramanujan_norms(x, 2, 40).shape == (0, 41)


def test_small_to_large_cap_retry_and_empty(eng):
    x = multi_sinusoid_batch(60, 3, 500)
    counts, per, pw, bs, st = eng.small_to_large(x, 0.001, cap=2)  # low threshold: many accepts
    for w in range(3):
        rper, rpw, rbs = po.small_to_large(x[w], 0.001)
        assert counts[w] == len(rper) and list(per[w, : counts[w]]) == rper
        assert rel_err(pw[w, : counts[w]], rpw) < TOL
        assert rel_err(bs[w, : counts[w]], np.array(rbs)) < TOL
    counts, per, pw, bs, st = eng.small_to_large(x, 10.0)  # nothing can pass
    assert not counts.any()


# ------------------------------------------------------------------------------ best_correlation
def test_best_correlation_matches_reference_golden(eng, golden):
    g = golden("best_correlation")
    x = multi_sinusoid_batch(2, 2, 700)
    per, nr, bs, st = eng.best_correlation(x, 5, None, 0.01)
    for k, w in enumerate((2, 3)):
        assert np.array_equal(per[k], g[f"bc_n700_w{w}_periods"])
        assert rel_err(nr[k], g[f"bc_n700_w{w}_norms"]) < TOL
        assert rel_err(bs[k], g[f"bc_n700_w{w}_bases"]) < TOL
    per, nr, bs, st = eng.best_correlation(multi_sinusoid_window(1, 4096)[None, :], 3)
    assert np.array_equal(per[0], g["bc_n4096_periods"])
    assert rel_err(nr[0], g["bc_n4096_norms"]) < TOL and rel_err(bs[0], g["bc_n4096_bases"]) < TOL


def test_best_frequency_matches_reference_golden(eng, golden):
    from pyperiod_amd import Periods

    g = golden("best_correlation")
    per, pw, bs = Periods().best_frequency(readme_window(2000, 0), None, 4)
    assert np.array_equal(per, g["bf_c1_periods"])
    assert rel_err(pw, g["bf_c1_powers"]) < TOL and rel_err(bs, g["bf_c1_bases"]) < TOL


def test_best_frequency_device_variants(eng):
    """Spectral peak on the device (direct DFT): window sizes other than N, flag combinations, a
    batch, float32, and the reference's division by zero when the peak is the DC bin."""
    from pyperiod_amd import Periods

    x = multi_sinusoid_batch(40, 3, 1500)
    for win in (None, 1500, 2048, 1000, 1499):
        per, pw, bs, st = eng.best_frequency(x, win, 3)
        assert not np.asarray(st).any()
        for w in range(3):
            rper, rpw, rbs = po.best_frequency(x[w], win, 3)
            assert np.array_equal(per[w], rper), (win, w)
            assert rel_err(pw[w], rpw) < TOL and rel_err(bs[w], rbs) < TOL, (win, w)
    # power-of-two win_size runs the in-LDS FFT, other sizes Bluestein's chirp convolution (direct DFT only when
    # neither fits the LDS): same peaks
    x4 = multi_sinusoid_batch(50, 3, 4096)
    for win in (None, 8192, 1024, 3000, 5000, 4095):  # FFT (powers of two) and Bluestein (the rest)
        per, pw, bs, st = eng.best_frequency(x4, win, 4)
        for w in range(3):
            rper, rpw, rbs = po.best_frequency(x4[w], win, 4)
            assert np.array_equal(per[w], rper), (win, w)
            assert rel_err(pw[w], rpw) < TOL and rel_err(bs[w], rbs) < TOL, (win, w)
    for trunc, orth in ((True, False), (False, True), (True, True)):
        per, pw, bs, st = eng.best_frequency(x[:1], None, 2, trunc, orth)
        rper, rpw, rbs = po.best_frequency(x[0], None, 2, trunc, orth)
        assert np.array_equal(per[0], rper) and rel_err(pw[0], rpw) < TOL and rel_err(bs[0], rbs) < TOL
    per32, pw32, bs32, _ = eng.best_frequency(x.astype(np.float32), None, 2)
    per64, pw64, _, _ = eng.best_frequency(x, None, 2)
    assert np.array_equal(per32, per64) and rel_err(pw32, pw64) < 1e-4
    # a signal riding on a large offset peaks at bin 0: 2 * win / 0 -> OverflowError in the reference
    with pytest.raises(OverflowError):
        Periods().best_frequency(x[0] + 50.0, None, 2)
    with pytest.warns(UserWarning):
        Periods().best_frequency(x[0], 1000, 1)
    res = Periods().best_frequency(x, None, 2)  # batched class surface
    assert res[0].shape == (3, 2) and res[2].shape == (3, 2, 1500)


# ------------------------------------------------------------------------------ Ramanujan
def test_ramanujan_matches_reference_golden(eng, golden):
    from pyperiod_amd import RamanujanPeriods

    g = golden("ramanujan")
    rp = RamanujanPeriods()
    got = rp.find_periods(multi_sinusoid_window(0, 240), 2, 80)
    assert got.shape == (81,) and got[0] == 0 and got[1] == 0
    assert rel_err(got, g["norms_n240"]) < 1e-5
    assert rel_err(got, po.ramanujan_norms_folded(multi_sinusoid_window(0, 240), 2, 80)) < TOL
    got = rp.find_periods(multi_sinusoid_window(1, 8192), 2, 64)
    assert rel_err(got, g["norms_n8192_pmax64"]) < 1e-5
    got = rp.find_periods(multi_sinusoid_window(2, 1000))
    assert got.shape == g["norms_n1000_default"].shape and rel_err(got, g["norms_n1000_default"]) < 1e-5
    x = multi_sinusoid_window(0, 240)
    basis = rp.Cq_complete(12, 240)
    proj = RamanujanPeriods.project(x, basis)
    assert proj.dtype == np.float32 and rel_err(proj, po.ramanujan_project(x, basis)) < 1e-6


# ------------------------------------------------------------------------------ QOPeriods
def test_qoperiods_matches_reference_golden(eng, golden):
    from pyperiod_amd import QOPeriods

    g = golden("qoperiods")
    qo = QOPeriods()
    a, d = qo.get_subspaces([12, 18, 8, 5], 1024)
    assert list(d.values()) == list(g["dims_12_18_8_5_vals"])
    w, rec = qo._solve_structured(g["solve_x"], a, d)
    assert rel_err(w, g["solve_w"]) < 1e-9 and rel_err(rec, g["solve_recon"]) < 1e-9
    w2, rec2 = QOPeriods.solve_quadratic(g["solve_x"], a)
    assert rel_err(w2, g["solve_w"]) < 1e-8 and rel_err(rec2, g["solve_recon"]) < 1e-8
    for tag, sig, kw in (
        ("c1", readme_window(2000, 0), dict(num=2, thresh=0.05)),
        ("w5", multi_sinusoid_window(5, 1536), dict(num=4, thresh=0.2, min_length=4, max_length=200)),
    ):
        out, res = qo.find_periods(sig, **kw)
        assert np.array_equal(out["periods"], g[f"fp_{tag}_periods"]), tag
        assert rel_err(out["norms"], g[f"fp_{tag}_norms"]) < TOL
        assert [int(k) for k in out["basis_dictionary"]] == list(g[f"fp_{tag}_dict_keys"])
        assert list(out["basis_dictionary"].values()) == list(g[f"fp_{tag}_dict_vals"])
        assert rel_err(out["weights"], g[f"fp_{tag}_weights"]) < 1e-8
        assert rel_err(res, g[f"fp_{tag}_residual"]) < 1e-8


def test_qoperiods_device_loop_batch_vs_oracle(eng, golden):
    """ph_qo_find_periods: the whole greedy loop on the device, a batch of windows per launch."""
    g = golden("qoperiods")
    x = multi_sinusoid_batch(5, 1, 1536)
    per, nrm, keeps, counts, wts, resid, st = eng.qo_find_periods(x, 4, 0.2, 4, 200)
    assert st[0] == 0
    nrep, nb = counts[0]
    assert np.array_equal(per[0, :nrep], g["fp_w5_periods"]) and rel_err(nrm[0, :nrep], g["fp_w5_norms"]) < TOL
    assert list(keeps[0, :nb]) == list(g["fp_w5_dict_vals"]) and list(per[0, :nb]) == list(g["fp_w5_dict_keys"])
    k = int(keeps[0, :nb].sum())
    assert rel_err(wts[0, :k], g["fp_w5_weights"]) < 1e-8 and rel_err(resid[0], g["fp_w5_residual"]) < 1e-8
    xb = multi_sinusoid_batch(70, 6, 900)
    per, nrm, keeps, counts, wts, resid, st = eng.qo_find_periods(xb, 5, 0.1, 2, None)
    for w in range(6):
        out, res = po.qo_find_periods(xb[w], 5, 0.1)
        nrep, nb = counts[w]
        assert np.array_equal(per[w, :nrep], out["periods"]), w
        assert rel_err(nrm[w, :nrep], out["norms"]) < TOL
        assert list(keeps[w, :nb]) == list(out["basis_dictionary"].values())
        k = int(keeps[w, :nb].sum())
        assert rel_err(wts[w, :k], out["weights"]) < 1e-8 and rel_err(resid[w], res) < 1e-8


def test_orthogonal_period_powers_match_reference_golden(eng, golden):
    from pyperiod_amd import QOPeriods

    g = golden("orth_powers")
    qo = QOPeriods()
    for tag, sig, max_p in (
        ("w1_n600", multi_sinusoid_window(1, 600), 200),
        ("w2_n1000", multi_sinusoid_window(2, 1000), None),
        ("c1", readme_window(2000, 0), 400),
    ):
        assert rel_err(qo.get_best_period_orthogonal(sig, max_p, False, True), g[f"pows_{tag}"]) < TOL
        assert rel_err(qo.get_best_period_orthogonal(sig, max_p, True, True), g[f"pows_norm_{tag}"]) < TOL
        assert qo.get_best_period_orthogonal(sig, max_p, True) == int(g[f"best_{tag}"])
        assert qo.get_best_period_orthogonal(sig, max_p) == int(g[f"best_raw_{tag}"])
    sig = multi_sinusoid_window(1, 600)
    assert rel_err([qo.eq_3(sig, q) for q in (1, 2, 7, 59)], g["eq3_w1_n600"][[0, 1, 6, 58]]) < 1e-12
    assert rel_err([qo.auto_corr(sig, k) for k in (0, 7, 595)], g["autocorr_w1_n600"][[0, 1, 85]]) < 1e-12
    xb = multi_sinusoid_batch(30, 5, 777)
    pows = eng.orth_powers(xb, 300, True)
    for w in range(5):
        assert rel_err(pows[w], po.orth_powers(xb[w], 300, True)) < TOL


# ------------------------------------------------------------------------------ class surface
def test_class_surface_matches_reference_behaviour(eng):
    from pyperiod_amd import Periods

    assert np.array_equal(Periods.project(np.arange(10.0), 3), [4.5, 4, 5, 4.5, 4, 5, 4.5, 4, 5, 4.5])
    assert np.array_equal(Periods.project(np.arange(10.0), 3, True), [3, 4, 5, 3, 4, 5, 3, 4, 5, 3])
    assert np.array_equal(Periods.project(np.arange(10.0), 3, return_single_period=True), [4.5, 4, 5])
    assert abs(Periods.periodic_norm(np.arange(10.0)) - 5.338539126015656) < 1e-14
    assert abs(Periods.periodic_norm(np.arange(10.0), 3) - 3.0822070014844885) < 1e-14
    with pytest.raises(AttributeError):  # list input, Periods.py:171
        Periods.project([1.0, 2.0, 3.0], 2)
    with pytest.raises(ValueError):  # 2-D input, Periods.py:176
        Periods.project(np.zeros((2, 8)), 2)
    x = multi_sinusoid_window(0, 64)
    keep = x.copy()
    Periods.project(x, 5, False, True)
    assert np.array_equal(x, keep)  # inputs are never mutated
    p = Periods(True, False)
    assert p.trunc_to_integer_multiple == (True, False)  # tuple quirk, Periods.py:610-611
    with pytest.warns(UserWarning):
        Periods(False, True).m_best(multi_sinusoid_window(1, 300), num=2)


# ------------------------------------------------------------------------------ full-size properties
def test_config2_full_size_properties(eng):
    """BASELINE config 2 shape (1024 windows x N=4096, m_best num=10): properties that do not
    need the oracle at full size, plus oracle spot checks on a few windows."""
    from pyperiod_amd import _ffi

    W, n, num = 1024, 4096, 10
    x = multi_sinusoid_batch(0, W, n)
    per, pw, bs, st = eng.m_best(x, num)
    assert not st.any()
    assert per.min() >= 2 and per.max() <= n // 3
    # every basis row is exactly periodic with its own period (tile structure)
    for w in (0, 17, 511, 1023):
        for k in range(num):
            p = int(per[w, k])
            assert np.array_equal(bs[w, k, p:], bs[w, k, :-p]), (w, k)
    # the batch result equals the per-window result (windows are independent)
    for w in (5, 900):
        p1, w1, b1, _ = eng.m_best(x[w : w + 1], num)
        assert np.array_equal(p1[0], per[w]) and np.array_equal(w1[0], pw[w]) and np.array_equal(b1[0], bs[w])
    # oracle spot check on two windows of the full batch
    for w in (2, 3):
        rper, rpw, rbs = po.m_best(x[w], num)
        assert np.array_equal(per[w], rper) and rel_err(pw[w], rpw) < TOL and rel_err(bs[w], rbs) < TOL
    # sweep: linearity in scale and Bessel's inequality ||P_p x|| <= ||x||
    sw = eng.sweep(x, 2, n // 3, _ffi.PH_SWEEP_NORM)
    nx = eng.periodic_norm(x)
    assert np.all(sw <= nx[:, None] * (1 + 1e-12))
    sw2 = eng.sweep(2.0 * x[:64], 2, n // 3, _ffi.PH_SWEEP_NORM)
    assert np.array_equal(sw2, 2.0 * sw[:64])  # power-of-two scaling is exact in binary fp
    # idempotence of the projection
    pl = [37, 64, 1365]
    pr = eng.project_batch(x[:8], pl)
    for k, p in enumerate(pl):
        again = eng.project_batch(pr[:, k, :].copy(), [p])[:, 0, :]
        assert rel_err(again, pr[:, k, :]) < 1e-14


def test_project_float32_trunc_keeps_float32_like_the_reference(golden):
    """Trunc mode on a float32 window: the reference's np.mean works on the float32 rectangle (Periods.py:178-184) and
    returns float32; the class surface runs the float kernels (row-order float32 sums, one division) instead of
    computing in fp64 and casting.  Fixture: the reference itself on float32 input (tests/golden/make_golden.py)."""
    from pyperiod_amd import Periods

    g = golden("project_f32")
    worst = 0.0
    for n in (97, 240, 4096):
        x = g[f"x_{n}"]
        assert x.dtype == np.float32
        for p in (2, 3, 7, 12, 64, 97, n // 2):
            if p > n:
                continue
            for orth in (False, True):
                want = g[f"n{n}_p{p}_o{int(orth)}"]
                got = Periods.project(x, p, True, orth)
                assert got.dtype == np.float32 and got.shape == want.shape
                if not orth:
                    assert np.array_equal(got, want), (n, p)  # the same float32 operations in the same order
                worst = max(worst, rel_err(got, want))
    assert worst < 5e-7  # orth: float32 subtractions of float32 projections, same order up to fused rounding


def test_sweep_ranges_that_cut_the_chains_of_short_periods(eng):
    """ph_sweep's norm modes take the periods up to 64 in chains L, L/2, L/4, ... (one row-split pass each,
    wave_chain_small, round 4): ranges that start inside a chain, end below 64, hold a single short period, or straddle 64
    must give the oracle's norms (Periods.py:501-510), for windows with strong short-period components and for lengths
    whose folds are ragged."""
    rng = np.random.default_rng(5)
    for n in (4096, 1000, 97):
        t = np.arange(n)
        x = np.stack([3.0 * rng.standard_normal(6)[t % 6] + 2.0 * rng.standard_normal(48)[t % 48] +
                      1.5 * rng.standard_normal(35)[t % 35] + 0.2 * rng.standard_normal(n) for _ in range(3)])
        for lo, hi in ((2, 64), (2, 63), (2, 50), (5, 64), (7, 40), (13, 13), (33, 100), (48, 96), (3, 24), (64, 200), (17, 300), (1, 70)):
            if hi > n:
                continue
            for mode in (0, 1):
                got = eng.sweep(x, lo, hi, mode)
                for w in range(3):
                    want = po.sweep_norms(x[w], lo, hi, gamma=(mode == 1))
                    s = slice(1, None) if lo == 1 else slice(None)  # p = 1: numpy sums the contiguous axis pairwise
                    assert elem_err(got[w][s], want[s]) < TOL, (n, lo, hi, mode, w)
                    if lo == 1:
                        assert abs(got[w][0] - want[0]) <= 1e-13 * max(1.0, abs(want[0]))
