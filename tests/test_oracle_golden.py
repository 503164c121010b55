"""Pin the CPU oracle (oracle/period_oracle.py) to golden vectors produced by the reference
itself (tests/golden/make_golden.py) and to the known answers of SURVEY.md section 4.
CPU only."""

import warnings

import numpy as np
import pytest

from conftest import elem_err, rel_err
from oracle import period_oracle as po
from pyperiod_amd.synth import multi_sinusoid_window, readme_window

FLAGS = [(False, False), (True, False), (False, True), (True, True)]
TOL = 1e-10  # BASELINE.json: fp64 powers/bases within 1e-10 relative


@pytest.fixture(autouse=True)
def _quiet():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def test_known_answers(golden):
    kat = golden("kat")
    assert np.array_equal(po.project(np.arange(10.0), 3), [4.5, 4, 5, 4.5, 4, 5, 4.5, 4, 5, 4.5])
    assert np.array_equal(po.project(np.arange(10.0), 3, True), [3, 4, 5, 3, 4, 5, 3, 4, 5, 3])
    assert np.array_equal(po.project(np.arange(10.0), 3), kat["project_arange10_p3"])
    assert np.array_equal(po.project(np.arange(10.0), 3, True), kat["project_arange10_p3_trunc"])
    assert po.periodic_norm(np.arange(10.0)) == 5.338539126015656 == kat["norm_arange10"]
    assert po.periodic_norm(np.arange(10.0), 3) == 3.0822070014844885 == kat["norm_arange10_p3"]
    assert po.phi(9) == 6 and po.phi(10) == 4
    assert list(kat["phi_9_10"]) == [6, 4]
    assert len(po.PRIMES) == 1229 == int(kat["n_primes_10000"])
    assert np.array_equal(np.rint(kat["cq6"]), [2, 1, -1, -2, -1, 1])
    assert np.array_equal(po.ramanujan_cq_exact(6), [2, 1, -1, -2, -1, 1])


def test_factor_set_iteration_order(golden):
    kat = golden("kat")
    off = kat["factor_order_off"]
    for k, n in enumerate(kat["factor_order_n"]):
        want = list(kat["factor_order_flat"][off[k] : off[k + 1]])
        assert [int(v) for v in po.factor_set(int(n), True)] == want, n


def test_project_bit_exact(golden):
    g = golden("project")
    for n in (10, 97, 240, 4096):
        x = g[f"x_{n}"]
        for p in (2, 3, 7, 12, 64, 97, n // 2):
            if p > n or p < 2:
                continue
            for trunc, orth in FLAGS:
                key = f"n{n}_p{p}_t{int(trunc)}_o{int(orth)}"
                got = po.project(x, p, trunc, orth)
                if n <= 240:
                    assert np.array_equal(got, g[key]), key
                else:
                    assert np.array_equal(got[:p], g[key + "_single"]), key
                    assert np.array_equal(got, np.tile(g[key + "_single"], n // p + 1)[:n])
                assert np.array_equal(po.project(x, p, trunc, orth, True), got[:p])


def test_project_is_row_order_accumulation(golden):
    """The spec the bit-exact HIP kernel is written to: p sequential accumulators over rows
    r = 0..R-1, then one division."""
    g = golden("project")
    x = g["x_4096"]
    n = x.size
    for p in (3, 7, 64, 97, 1000, 2048):
        acc = np.zeros(p)
        for r in range(-(-n // p)):
            seg = x[r * p : (r + 1) * p]
            acc[: seg.size] += seg
        assert np.array_equal(acc, po.fold_sums(x, p))
        assert np.array_equal(acc / po.fold_counts(n, p), po.project(x, p, return_single_period=True))


def test_project_input_untouched_and_edge_cases():
    x = multi_sinusoid_window(0, 50)
    keep = x.copy()
    po.project(x, 7, False, True)
    assert np.array_equal(x, keep)
    # p > N: the visible part is the signal itself (counts are 1), Periods.py:188-198
    assert np.array_equal(po.project(x, 64), x)
    # prime p with orthogonalize: nothing removed (DC stays), Periods.py:208-214
    assert np.array_equal(po.project(x, 7, False, True), po.project(x, 7, False, False))


def test_sweeps(golden):
    g = golden("sweep")
    n = 4096
    for w in range(4):
        x = multi_sinusoid_window(w, n)
        assert rel_err(po.sweep_norms(x, 2, n // 3), g[f"plain_w{w}"]) < 1e-13
        assert rel_err(po.sweep_norms(x, 2, n // 3, gamma=True), g[f"gamma_w{w}"]) < 1e-13
    x = multi_sinusoid_window(0, n)
    assert np.array_equal(po.sweep_maxabs(x, 2, n // 3), g["maxabs_w0"])
    for trunc, orth in FLAGS[1:]:
        want = g[f"plain_w0_t{int(trunc)}_o{int(orth)}"]
        assert rel_err(po.sweep_norms(x, 2, n // 3, trunc=trunc, orth=orth), want) < 1e-13


def test_norm_identities():
    """Identities the fast kernels rely on, non-trunc / non-orth only."""
    x = multi_sinusoid_window(3, 4096)
    n = x.size
    for p in (2, 37, 100, 1365, 2048):
        s = po.fold_sums(x, p)
        c = po.fold_counts(n, p)
        proj = po.project(x, p)
        assert abs(np.sum(s * s / c) - proj @ proj) <= 1e-13 * (proj @ proj)
        lhs = (x - proj) @ (x - proj)
        assert abs(lhs - (x @ x - proj @ proj)) <= 1e-12 * (x @ x)


def test_small_to_large(golden):
    g = golden("small_to_large")
    per, pw, bs = po.small_to_large(readme_window(2000, 0), thresh=0.1)
    assert per == list(g["c1_periods"])
    assert rel_err(pw, g["c1_powers"]) < TOL and rel_err(np.array(bs), g["c1_bases"]) < TOL
    for w in range(4):
        per, pw, bs = po.small_to_large(multi_sinusoid_window(w, 4096), thresh=0.05)
        assert per == list(g[f"w{w}_periods"]), w
        assert rel_err(pw, g[f"w{w}_powers"]) < TOL
        if w == 1:
            assert rel_err(np.array(bs), g["w1_bases"]) < TOL
    for trunc, orth in FLAGS[1:]:
        tag = f"n1200_t{int(trunc)}_o{int(orth)}"
        per, pw, bs = po.small_to_large(multi_sinusoid_window(2, 1200), 0.05, None, trunc, orth)
        assert per == list(g[tag + "_periods"]), tag
        assert rel_err(pw, g[tag + "_powers"]) < TOL
        assert rel_err(np.array(bs).reshape(len(per), 1200), g[tag + "_bases"]) < TOL
    per, pw, _ = po.small_to_large(multi_sinusoid_window(3, 600), thresh=0.02, n_periods=100)
    assert per == list(g["n600_np100_periods"]) and rel_err(pw, g["n600_np100_powers"]) < TOL


@pytest.mark.parametrize("name,gamma", [("m_best", False), ("m_best_gamma", True)])
def test_m_best(golden, name, gamma):
    g = golden("m_best")
    for w in range(2):  # 2.3 s per window on the oracle; windows 2,3 are used by the GPU tests
        per, pw, bs = po.m_best(multi_sinusoid_window(w, 4096), num=10, gamma=gamma)
        assert per.dtype == np.uint32 and np.array_equal(per, g[f"{name}_w{w}_periods"]), w
        assert rel_err(pw, g[f"{name}_w{w}_powers"]) < TOL
        if w == 1:
            assert rel_err(bs, g[f"{name}_w1_bases"]) < TOL
    for w in (4, 5):
        per, pw, bs = po.m_best(multi_sinusoid_window(w, 1500), 6, 300, 3, gamma)
        assert np.array_equal(per, g[f"{name}_n1500_w{w}_periods"])
        assert rel_err(pw, g[f"{name}_n1500_w{w}_powers"]) < TOL
        assert rel_err(bs, g[f"{name}_n1500_w{w}_bases"]) < TOL
    per, pw, bs = po.m_best(readme_window(2000, 0), num=10, gamma=gamma)
    assert np.array_equal(per, g[f"{name}_c1_periods"])
    assert rel_err(pw, g[f"{name}_c1_powers"]) < TOL and rel_err(bs, g[f"{name}_c1_bases"]) < TOL
    for trunc, orth in FLAGS[1:]:
        tag = f"{name}_n900_t{int(trunc)}_o{int(orth)}"
        per, pw, bs = po.m_best(multi_sinusoid_window(6, 900), 5, None, 2, gamma, trunc, orth)
        assert np.array_equal(per, g[tag + "_periods"]), tag
        assert rel_err(pw, g[tag + "_powers"]) < TOL and rel_err(bs, g[tag + "_bases"]) < TOL


def test_best_correlation_and_frequency(golden):
    g = golden("best_correlation")
    for w in (2, 3):
        per, nr, bs = po.best_correlation(multi_sinusoid_window(w, 700), num=5, ratio=0.01)
        assert np.array_equal(per, g[f"bc_n700_w{w}_periods"])
        assert rel_err(nr, g[f"bc_n700_w{w}_norms"]) < TOL
        assert rel_err(bs, g[f"bc_n700_w{w}_bases"]) < TOL
    per, nr, bs = po.best_correlation(multi_sinusoid_window(1, 4096), num=3)
    assert np.array_equal(per, g["bc_n4096_periods"])
    assert rel_err(nr, g["bc_n4096_norms"]) < TOL and rel_err(bs, g["bc_n4096_bases"]) < TOL
    per, pw, bs = po.best_frequency(readme_window(2000, 0), None, 4)
    assert np.array_equal(per, g["bf_c1_periods"])
    assert rel_err(pw, g["bf_c1_powers"]) < TOL and rel_err(bs, g["bf_c1_bases"]) < TOL


def test_ramanujan(golden):
    g = golden("ramanujan")
    for q in range(1, 65):
        assert rel_err(po.ramanujan_cq(q), g[f"cq_{q}"]) < 1e-12
        assert np.array_equal(np.rint(g[f"cq_{q}"]).astype(np.int64), po.ramanujan_cq_exact(q))
        assert np.max(np.abs(g[f"cq_{q}"] - np.rint(g[f"cq_{q}"]))) < 1e-10
    assert rel_err(po.ramanujan_dictionary(6, 20), g["cq_complete_6_20"]) < 1e-12
    x = multi_sinusoid_window(0, 240)
    assert rel_err(po.ramanujan_find_periods(x, 2, 80), g["norms_n240"]) < 1e-6
    # the fp64 folded form the kernel evaluates stays within the reference's own float32 noise
    got = po.ramanujan_norms_folded(x, 2, 80)
    assert rel_err(got, g["norms_n240"]) < 1e-5
    x = multi_sinusoid_window(1, 8192)
    assert rel_err(po.ramanujan_norms_folded(x, 2, 64), g["norms_n8192_pmax64"]) < 1e-5
    x = multi_sinusoid_window(2, 1000)
    want = g["norms_n1000_default"]
    assert want.shape == (1000 // 3 + 1,) and want[0] == 0 and want[1] == 0
    assert rel_err(po.ramanujan_norms_folded(x), want) < 1e-5


def test_qoperiods_pieces(golden):
    g = golden("qoperiods")
    dims = po.qo_subspace_dims([37, 64, 101], 16384)
    assert [int(k) for k in dims] == list(g["dims_37_64_101_keys"])
    assert list(dims.values()) == list(g["dims_37_64_101_vals"]) == [37, 63, 100]
    assert list(po.qo_subspace_dims([12, 18, 8, 5], 1024).values()) == list(g["dims_12_18_8_5_vals"])
    assert np.array_equal(po.qo_natural_rows(5, 12, 3), g["pp_5_12_keep3"])
    a, _ = po.qo_get_subspaces([12, 18, 8, 5], 1024)
    w, rec = po.qo_solve_quadratic(g["solve_x"], a)
    assert rel_err(w, g["solve_w"]) < 1e-9 and rel_err(rec, g["solve_recon"]) < 1e-9
    for tag, sig, kw in (
        ("c1", readme_window(2000, 0), dict(num=2, thresh=0.05)),
        ("w5", multi_sinusoid_window(5, 1536), dict(num=4, thresh=0.2, min_length=4, max_length=200)),
    ):
        out, res = po.qo_find_periods(sig, **kw)
        assert np.array_equal(out["periods"], g[f"fp_{tag}_periods"]), tag
        assert rel_err(out["norms"], g[f"fp_{tag}_norms"]) < TOL
        assert [int(k) for k in out["basis_dictionary"]] == list(g[f"fp_{tag}_dict_keys"])
        assert list(out["basis_dictionary"].values()) == list(g[f"fp_{tag}_dict_vals"])
        assert rel_err(out["weights"], g[f"fp_{tag}_weights"]) < 1e-8
        assert rel_err(res, g[f"fp_{tag}_residual"]) < 1e-8


def test_orthogonal_period_powers(golden):
    g = golden("orth_powers")
    for tag, sig, max_p in (
        ("w1_n600", multi_sinusoid_window(1, 600), 200),
        ("w2_n1000", multi_sinusoid_window(2, 1000), None),
        ("c1", readme_window(2000, 0), 400),
    ):
        assert rel_err(po.orth_powers(sig, max_p), g[f"pows_{tag}"]) < TOL
        assert rel_err(po.orth_powers(sig, max_p, True), g[f"pows_norm_{tag}"]) < TOL
        assert po.best_period_orthogonal(sig, max_p, True) == int(g[f"best_{tag}"])
        assert po.best_period_orthogonal(sig, max_p) == int(g[f"best_raw_{tag}"])
    sig = multi_sinusoid_window(1, 600)
    assert rel_err([po.eq_3(sig, q) for q in range(1, 60)], g["eq3_w1_n600"]) < 1e-12
    assert rel_err([po.auto_corr(sig, k) for k in range(0, 600, 7)], g["autocorr_w1_n600"]) < 1e-13



# ---------------------------------------------------------------------------------- round-2 fixtures
@pytest.mark.parametrize("name,gamma", [("m_best_gamma", True), ("m_best", False)])
def test_m_best_step2_splits(golden, name, gamma):
    """Calls in which step 2 really splits a row (np.insert fires, Periods.py:581-594); in gamma mode
    this pins the stale-`p` divisor of Periods.py:559,572."""
    g = golden("m_best_split")
    for n, ml, num, w in g["cases"]:
        tag = f"{name}_n{n}_ml{ml}_num{num}_w{w}"
        if gamma:
            assert int(g[tag + "_splits"]) >= 1
        per, pw, bs = po.m_best(multi_sinusoid_window(int(w), int(n)), int(num), int(ml), 2, gamma)
        assert np.array_equal(per, g[tag + "_periods"]), tag
        assert rel_err(pw, g[tag + "_powers"]) < TOL and elem_err(pw, g[tag + "_powers"]) < 1e-9
        assert rel_err(bs, g[tag + "_bases"]) < TOL


@pytest.mark.parametrize("name,gamma", [("m_best", False), ("m_best_gamma", True)])
def test_m_best_periods_far_above_a_third_of_the_window(golden, name, gamma):
    """max_length up to 0.88 N: periods beyond N/2 (single-sample residues) and 2N/3; the first case splits
    454 -> 227 in step 2 (Periods.py:581-594)."""
    g = golden("m_best_large_p")
    for n, ml, num, _ in g["cases"]:
        tag = f"{name}_n{n}_ml{ml}_num{num}"
        per, pw, bs = po.m_best(g[f"x_n{n}"], int(num), int(ml), 2, gamma)
        assert np.array_equal(per, g[tag + "_periods"]), tag
        assert rel_err(pw, g[tag + "_powers"]) < TOL and rel_err(bs, g[tag + "_bases"]) < TOL, tag


def test_ramanujan_config3_shape(golden):
    """N = 8192, q = 2..512 (BASELINE config 3): the fp64 folded form against the float32 reference."""
    g = golden("ramanujan_c3")
    for w in (0, 1):
        want = g[f"norms_n8192_pmax512_w{w}"]
        got = po.ramanujan_norms_folded(multi_sinusoid_window(w, 8192), 2, 512)
        assert want.shape == (513,) and rel_err(got, want) < 1e-5
        assert elem_err(got, want, 1e-4) < 1e-4  # weak subspaces sit in the reference's float32 noise


def test_ramanujan_find_periods_with_weights(golden):
    g = golden("ramanujan_weights")
    for tag, sig, kw in (
        ("n240", multi_sinusoid_window(0, 240), dict(min_length=2, max_length=80, thresh=0.2)),
        ("n1000", multi_sinusoid_window(2, 1000), dict(thresh=0.3)),
        ("n600", multi_sinusoid_window(5, 600), dict(min_length=3, max_length=150, thresh=0.1)),
    ):
        out, res = po.ramanujan_find_periods_with_weights(sig, **kw)
        assert np.array_equal(out["periods"], g[f"{tag}_periods"]), tag
        assert rel_err(out["norms"], g[f"{tag}_norms"]) < 1e-5
        assert [int(k) for k in out["basis_dictionary"]] == list(g[f"{tag}_dict_keys"])
        assert list(out["basis_dictionary"].values()) == list(g[f"{tag}_dict_vals"])
        assert rel_err(out["weights"], g[f"{tag}_weights"]) < 1e-8 and rel_err(res, g[f"{tag}_residual"]) < 1e-8


def test_qoperiods_config5_length(golden):
    """QOPeriods.find_periods on fp32-rounded windows of N = 16384 (config 5's length)."""
    g = golden("qoperiods_c5")
    for tag, w, kw in (
        ("w0", 0, dict(num=3, thresh=0.1, min_length=8, max_length=300)),
        ("w7", 7, dict(num=4, thresh=0.05, min_length=8, max_length=300)),
    ):
        sig = multi_sinusoid_window(w, 16384, dtype=np.float32).astype(np.float64)
        out, res = po.qo_find_periods(sig, **kw)
        assert np.array_equal(out["periods"], g[f"fp_{tag}_periods"]), tag
        assert rel_err(out["norms"], g[f"fp_{tag}_norms"]) < TOL
        assert list(out["basis_dictionary"].values()) == list(g[f"fp_{tag}_dict_vals"])
        assert rel_err(out["weights"], g[f"fp_{tag}_weights"]) < 1e-7
        assert rel_err(res, g[f"fp_{tag}_residual"]) < 1e-6  # the fixture stores the residual as float32


def test_elementwise_bar_on_sweeps(golden):
    """north_star's 1e-10 is a relative bar: hold the oracle to it entry by entry."""
    g = golden("sweep")
    x = multi_sinusoid_window(0, 4096)
    assert elem_err(po.sweep_norms(x, 2, 4096 // 3), g["plain_w0"]) < 1e-12
    assert elem_err(po.sweep_norms(x, 2, 4096 // 3, gamma=True), g["gamma_w0"]) < 1e-12


def test_oracle_project_float32_trunc_matches_reference(golden):
    """float32 windows in trunc mode stay float32 (np.mean on the float32 rectangle, Periods.py:178-184)."""
    g = golden("project_f32")
    for n in (97, 240, 4096):
        x = g[f"x_{n}"]
        for p in (2, 3, 7, 12, 64, 97, n // 2):
            if p > n:
                continue
            for orth in (False, True):
                got = po.project(x, p, True, orth)
                assert got.dtype == np.float32 and np.array_equal(got, g[f"n{n}_p{p}_o{int(orth)}"]), (n, p, orth)
