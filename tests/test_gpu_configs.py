"""GPU tests at the full sizes of BASELINE.json configs 3, 4 and 5 (config 2 is covered in
test_gpu_parity.py): size-independent properties on the whole batch plus oracle / golden spot
checks on a few windows, and the edge cases of the batch interface."""

import warnings

import numpy as np
import pytest

from conftest import elem_err, rel_err
from oracle import period_oracle as po
from pyperiod_amd.synth import multi_sinusoid_batch, multi_sinusoid_window

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


def test_config4_shard_small_to_large(eng, golden):
    """Config 4, one GPU's shard shape: windows x N=4096, small_to_large(thresh=0.05)."""
    W, n, thresh = 2048, 4096, 0.05
    x = multi_sinusoid_batch(0, W, n)
    counts, per, pw, bs, st = eng.small_to_large(x, thresh, cap=40, want_bases=False)
    assert not st.any() and counts.min() >= 1
    g = golden("small_to_large")
    for w in range(4):  # the reference's own answers for windows 0..3
        assert list(per[w, : counts[w]]) == list(g[f"w{w}_periods"])
        assert rel_err(pw[w, : counts[w]], g[f"w{w}_powers"]) < TOL
    for w in range(W):
        k = counts[w]
        assert np.all(np.diff(per[w, :k]) > 0) and per[w, 0] >= 2 and per[w, k - 1] <= n // 2
        assert np.all(pw[w, :k] > thresh) and not per[w, k:].any()
    # energy bookkeeping: the accepted norm drops telescope to (||x|| - ||residual||) / ||x||
    sub = x[:64]
    c2, p2, w2, b2, _ = eng.small_to_large(sub, thresh, cap=40, want_bases=True)
    assert np.array_equal(c2, counts[:64]) and np.array_equal(p2, per[:64])
    for w in range(64):
        k = c2[w]
        resid = sub[w] - b2[w, :k].sum(0)
        total = (po.periodic_norm(sub[w]) - po.periodic_norm(resid)) / po.periodic_norm(sub[w])
        assert abs(w2[w, :k].sum() - total) < 1e-9
        for i in range(k):  # each base is exactly periodic with its period
            p = int(p2[w, i])
            assert np.array_equal(b2[w, i, p:], b2[w, i, :-p])
    # oracle spot check beyond the golden windows
    for w in (100, 2047):
        rper, rpw, _ = po.small_to_large(x[w], thresh)
        assert list(per[w, : counts[w]]) == rper and rel_err(pw[w, : counts[w]], rpw) < TOL


def test_config4_full_batch_device_resident(eng):
    """BASELINE config 4 in one call: 65 536 windows x N=4096 resident in HBM (2 GiB), periods only.
    Index widths, grid size and the per-window independence at full scale."""
    import torch

    base = multi_sinusoid_batch(0, 256, 4096)
    x = torch.from_numpy(base).cuda().repeat(256, 1)
    counts, per, pw, bs, st = eng.small_to_large(x, 0.05, None, False, False, cap=24, want_bases=False)
    counts, per, pw = counts.cpu().numpy(), per.cpu().numpy(), pw.cpu().numpy()
    assert int(st.abs().max()) == 0 and counts.shape == (65536,)
    # every copy of a window gives the same answer, and it is the oracle's
    assert np.array_equal(counts.reshape(256, 256), np.broadcast_to(counts[:256], (256, 256)))
    assert np.array_equal(per.reshape(256, 256, -1), np.broadcast_to(per[:256], (256, 256, per.shape[1])))
    for w in (5, 200):
        rper, rpw, _ = po.small_to_large(base[w], 0.05)
        for k in (w, w + 256 * 255):
            assert list(per[k, : counts[k]]) == rper and rel_err(pw[k, : counts[k]], rpw) < TOL
    out = eng.m_best(x[:16384].contiguous(), 10)
    pp = out[0].cpu().numpy().view(np.uint32)
    assert np.array_equal(pp.reshape(64, 256, 10), np.broadcast_to(pp[:256], (64, 256, 10)))
    assert np.array_equal(pp[7 + 256 * 63], po.m_best(base[7], 10)[0])


def test_small_to_large_threshold_on_a_knife_edge(eng):
    """The screen of k_small_to_large skips a period only when its estimate plus a rigorous bound on
    its error stays below the threshold.  Plant the threshold within 1e-12 of the
    drop of an accepted period, on both sides, at N = 16384 (the bound grows with N) and N = 4096:
    the period list must flip exactly where the oracle's does."""
    for n, w in ((16384, 3), (4096, 9)):
        x = multi_sinusoid_window(w, n)
        rper, rpw, _ = po.small_to_large(x, 0.02)
        assert len(rper) >= 3
        for k in (0, len(rper) // 2, len(rper) - 1):
            for thresh in (rpw[k] - 1e-12, rpw[k] + 1e-12):
                want_per, want_pw, _ = po.small_to_large(x, thresh)
                counts, per, pw, _, st = eng.small_to_large(x[None, :], thresh, cap=64, want_bases=False)
                assert list(per[0, : counts[0]]) == want_per, (n, k, thresh)
                assert rel_err(pw[0, : counts[0]], want_pw) < TOL


def test_config3_ramanujan_batch(eng, golden):
    """BASELINE config 3 as stated: N=8192, Pmax=512, the whole 4096-window batch in one launch
    (device-resident), every q in [2, 512] checked."""
    import torch

    W, n, pmax = 4096, 8192, 512
    x = multi_sinusoid_batch(0, W, n)
    xd = torch.from_numpy(x).cuda()
    out = eng.ramanujan_norms(xd, 2, pmax).cpu().numpy()
    assert out.shape == (W, pmax + 1) and not out[:, :2].any() and np.all(out[:, 2:] > 0)
    g = golden("ramanujan")
    assert rel_err(out[1, :65], g["norms_n8192_pmax64"]) < 1e-5  # the reference's answer for window 1, q <= 64
    g3 = golden("ramanujan_c3")
    for w in (0, 1):  # the reference itself for q = 2 .. 512 (float32 accumulation: 1e-5)
        assert rel_err(out[w], g3[f"norms_n8192_pmax512_w{w}"]) < 1e-5
        assert elem_err(out[w], g3[f"norms_n8192_pmax512_w{w}"], 1e-4) < 1e-4
    for w in (7, 4095):  # the fp64 folded oracle, all 511 periods, entry by entry
        want = po.ramanujan_norms_folded(x[w], 2, pmax)
        assert rel_err(out[w], want) < TOL and elem_err(out[w], want) < 1e-9
    single = eng.ramanujan_norms(x[3:4], 2, pmax)
    assert np.array_equal(single[0], out[3])  # batch == per-window
    sub = eng.ramanujan_norms(x[3:4], 100, 300)  # a sub-range takes other roots: same values to rounding
    assert not sub[0, :100].any() and rel_err(sub[0, 100:], out[3, 100:301]) < 1e-12
    # exactness of the subspace split: sum over q | n of E_q energy == energy of the n-periodic part
    y = po.project(x[5][:8190], 90)  # a 90-periodic signal (8190 = 91 * 90)
    r = eng.ramanujan_norms(y[None, :], 1, 90)[0]
    divs = [d for d in range(1, 91) if 90 % d == 0]
    # orthogonality is exact only for q | N (otherwise the finite window leaks)
    others = [q for q in range(1, 91) if 90 % q and 8190 % q == 0]
    assert len(others) >= 10 and r[others].max() < 1e-18 * r[divs].max()


def test_ramanujan_without_the_pad_behind_the_window(eng):
    """Shapes where dropping the zeroed pad behind the LDS window makes room for one more wavefront (N = 5000 fp64 with
    q <= 640: 15 wavefronts with the pad, 16 without; config 3 is the other such shape): a fold then reads past the window
    into the strips for lanes whose sums it discards.  Odd batch, a ragged length, q_lo = 1 (slot 0 of the output row is
    the root queue and must come back as zero), every period against the fp64 folded oracle."""
    x = multi_sinusoid_batch(77, 3, 5000)
    out = eng.ramanujan_norms(x, 1, 640)
    assert out.shape == (3, 641) and not out[:, 0].any()
    for w in (0, 2):
        want = po.ramanujan_norms_folded(x[w], 1, 640)
        assert rel_err(out[w], want) < TOL and elem_err(out[w], want) < 1e-9
    again = eng.ramanujan_norms(x[1:2], 1, 640)
    assert np.array_equal(again[0], out[1])  # batch == per-window, and the queue slot was reset


def test_ramanujan_default_range(eng, golden):
    """RamanujanPeriods().find_periods(x) with the reference's default max_length = len(x) // 3
    (RamanujanPeriods.py:68-69) at N = 4096 .. 16384: the per-wavefront strips are sized by the range."""
    from pyperiod_amd import RamanujanPeriods

    rp = RamanujanPeriods()
    x = multi_sinusoid_window(3, 4096)
    got = rp.find_periods(x)
    want = golden("ramanujan_default")["norms_n4096_default_w3"]  # the reference, q = 2 .. 1365
    assert got.shape == want.shape == (4096 // 3 + 1,)
    assert rel_err(got, want) < 1e-5 and elem_err(got, want, 1e-4) < 1e-4
    fold = po.ramanujan_norms_folded(x)
    assert rel_err(got, fold) < TOL and elem_err(got, fold) < 1e-9
    for n in (8192, 16384):
        x = multi_sinusoid_window(4, n)
        got = rp.find_periods(x)
        assert got.shape == (n // 3 + 1,) and np.all(got[2:] > 0)
        qs = [2, 3, 63, 64, 65, 97, 360, 509, 1024, 1365, 2048, n // 6, n // 6 + 1, 2520, n // 3 - 1, n // 3]
        want = np.array([po.ramanujan_norm_folded_q(x, q) for q in qs])
        assert elem_err(got[qs], want) < 1e-9, n
    # fp32 windows and a batch take the same path
    xb = multi_sinusoid_batch(0, 3, 4096, dtype=np.float32)
    gb = eng.ramanujan_norms(xb)
    assert gb.shape == (3, 1366)
    assert rel_err(gb[1], po.ramanujan_norms_folded(xb[1].astype(np.float64))) < 1e-5


def test_ramanujan_find_periods_with_weights(eng, golden):
    """RamanujanPeriods.find_periods_with_weights (RamanujanPeriods.py:88-122) against the reference run
    with its two v1 defects repaired in the harness (tests/golden/make_golden.py, shim 4)."""
    from pyperiod_amd import RamanujanPeriods

    g = golden("ramanujan_weights")
    for tag, sig, kw in (
        ("n240", multi_sinusoid_window(0, 240), dict(min_length=2, max_length=80, thresh=0.2)),
        ("n1000", multi_sinusoid_window(2, 1000), dict(thresh=0.3)),
        ("n600", multi_sinusoid_window(5, 600), dict(min_length=3, max_length=150, thresh=0.1)),
    ):
        out, res = RamanujanPeriods().find_periods_with_weights(sig, **kw)
        assert np.array_equal(out["periods"], g[f"{tag}_periods"]), tag
        assert rel_err(out["norms"], g[f"{tag}_norms"]) < 1e-5
        assert [int(k) for k in out["basis_dictionary"]] == list(g[f"{tag}_dict_keys"])
        assert list(out["basis_dictionary"].values()) == list(g[f"{tag}_dict_vals"])
        assert out["subspaces"].shape == (int(g[f"{tag}_dict_vals"].sum()), sig.size)
        assert rel_err(out["weights"], g[f"{tag}_weights"]) < 1e-8 and rel_err(res, g[f"{tag}_residual"]) < 1e-8
    # a caller-supplied test function overrides thresh (RamanujanPeriods.py:93-94)
    out, res = RamanujanPeriods().find_periods_with_weights(multi_sinusoid_window(0, 240), 2, 80, test_function=lambda v: np.array([7, 12]))
    assert list(out["periods"]) == [7, 12] and list(out["basis_dictionary"].values()) == [7, 11]


def test_config5_fp32_blocks(eng):
    """Config 5 pieces on fp32 windows of N=16384: orthogonalised projection, A x folds and the
    A^T w reconstruction, against the fp64 oracle on the fp32-rounded inputs (1e-4 relative,
    build-defined -- the reference has no fp32 path)."""
    n = 16384
    x32 = multi_sinusoid_batch(10, 3, n, dtype=np.float32)
    x64 = x32.astype(np.float64)
    plist = [37, 64, 101, 1260]
    for trunc, orth in ((False, True), (True, True), (False, False)):
        out = eng.project_batch(x32, plist, trunc, orth)
        assert out.dtype == np.float32
        for w in range(3):
            for k, p in enumerate(plist):
                assert rel_err(out[w, k], po.project(x64[w], p, trunc, orth)) < 1e-4
    from pyperiod_amd import QOPeriods

    qo = QOPeriods()
    a, dims = qo.get_subspaces([37, 64, 101], n)
    p_list, keep = [int(k) for k in dims], list(dims.values())
    assert keep == [37, 63, 100]
    folds = eng.fold_sums(x32, p_list, keep)
    assert rel_err(folds, x64 @ a.T) < 1e-4
    gram = eng.fold_sums(a, p_list, keep)
    assert np.array_equal(gram, a @ a.T)  # integer co-occurrence counts: exact
    w = np.linalg.solve(gram, folds[0])
    rec = eng.tile_sum(w[None, :], n, p_list, keep)[0]
    assert rel_err(rec, a.T @ w) < 1e-12
    # least-squares property: the residual is orthogonal to every dictionary row
    resid = x64[0] - rec
    assert np.max(np.abs(a @ resid)) < 1e-3 * np.max(np.abs(folds[0]))
    # fp32 sweep against the fp64 oracle
    from pyperiod_amd import _ffi

    sw = eng.sweep(x32[:1], 2, 600, _ffi.PH_SWEEP_NORM_GAMMA)[0]
    assert rel_err(sw, po.sweep_norms(x64[0], 2, 600, gamma=True)) < 1e-4


def test_config5_device_loop_fp32_n16384(eng, golden):
    """Config 5's shape and dtype through ph_qo_find_periods: fp32 windows of N = 16384, the whole
    greedy loop on the device.  Against the reference's own answer on the rounded input (fixture) and
    the fp64 oracle: periods / rows kept exact, norms / weights / residual 1e-4 (build-defined fp32 bar)."""
    g = golden("qoperiods_c5")
    n = 16384
    for tag, w, kw in (
        ("w0", 0, dict(num=3, thresh=0.1, min_length=8, max_length=300)),
        ("w7", 7, dict(num=4, thresh=0.05, min_length=8, max_length=300)),
    ):
        x32 = multi_sinusoid_window(w, n, dtype=np.float32)[None, :]
        per, nrm, keeps, counts, wts, resid, st = eng.qo_find_periods(x32, kw["num"], kw["thresh"], kw["min_length"], kw["max_length"], 1024)
        assert st[0] == 0 and resid.dtype == np.float32
        nrep, nb = counts[0]
        assert np.array_equal(per[0, :nrep], g[f"fp_{tag}_periods"]), tag
        assert list(keeps[0, :nb]) == list(g[f"fp_{tag}_dict_vals"]) and list(per[0, :nb]) == list(g[f"fp_{tag}_dict_keys"])
        assert rel_err(nrm[0, :nrep], g[f"fp_{tag}_norms"]) < 1e-4
        k = int(keeps[0, :nb].sum())
        assert rel_err(wts[0, :k], g[f"fp_{tag}_weights"]) < 1e-4
        assert rel_err(resid[0], g[f"fp_{tag}_residual"]) < 1e-4
    # a batch of fp32 windows against the fp64 oracle on the rounded inputs
    xb = multi_sinusoid_batch(20, 4, n, dtype=np.float32)
    per, nrm, keeps, counts, wts, resid, st = eng.qo_find_periods(xb, 3, 0.1, 8, 300, 1024)
    for w in range(4):
        out, res = po.qo_find_periods(xb[w].astype(np.float64), 3, 0.1, 8, 300)
        nrep, nb = counts[w]
        assert st[w] == 0 and np.array_equal(per[w, :nrep], out["periods"]), w
        assert list(keeps[w, :nb]) == list(out["basis_dictionary"].values())
        k = int(keeps[w, :nb].sum())
        assert rel_err(nrm[w, :nrep], out["norms"]) < 1e-4
        assert rel_err(wts[w, :k], out["weights"]) < 1e-4 and rel_err(resid[w], res) < 1e-4
    # the same windows in fp64 meet the fp64 bar
    per64, nrm64, keeps64, counts64, wts64, resid64, st64 = eng.qo_find_periods(xb.astype(np.float64), 3, 0.1, 8, 300, 1024)
    for w in range(4):
        out, res = po.qo_find_periods(xb[w].astype(np.float64), 3, 0.1, 8, 300)
        nrep, nb = counts64[w]
        k = int(keeps64[w, :nb].sum())
        assert np.array_equal(per64[w, :nrep], out["periods"]) and rel_err(nrm64[w, :nrep], out["norms"]) < TOL
        assert rel_err(wts64[w, :k], out["weights"]) < 1e-8 and rel_err(resid64[w], res) < 1e-8


def test_qoperiods_falls_back_to_the_host_loop(eng):
    """Windows / dictionaries the single-launch kernel cannot hold: QOPeriods.find_periods
    must not raise -- the host-driven loop (ph_sweep / ph_fold_sums / ph_tile_sum + LAPACK) takes over."""
    from pyperiod_amd import QOPeriods

    # a dictionary of more rows than fits beside an N = 16384 fp64 window (three periods near 300)
    t = np.arange(16384, dtype=np.float64)
    sig = np.sin(2 * np.pi * t / 299.0) + 0.8 * np.sin(2 * np.pi * t / 293.0 + 1.0) + 0.6 * np.sin(2 * np.pi * t / 283.0 + 2.0)
    sig = sig + 0.01 * np.random.default_rng(5).standard_normal(t.size)
    out, res = QOPeriods().find_periods(sig, num=3, thresh=0.01, min_length=200, max_length=300)
    want, wres = po.qo_find_periods(sig, 3, 0.01, 200, 300)
    assert sum(out["basis_dictionary"].values()) > 512
    assert np.array_equal(out["periods"], want["periods"]) and rel_err(out["norms"], want["norms"]) < TOL
    assert list(out["basis_dictionary"].values()) == list(want["basis_dictionary"].values())
    assert rel_err(out["weights"], want["weights"]) < 1e-7 and rel_err(res, wres) < 1e-7


def test_qoperiods_host_driven_variants(eng):
    """The host-driven greedy loop (custom test function, update_weights=False) against the oracle /
    the single-launch kernel: same periods, norms, dictionary and weights."""
    from pyperiod_amd import QOPeriods
    from pyperiod_amd.Periods import rms

    sig = multi_sinusoid_window(5, 1536)
    dev_out, dev_res = QOPeriods().find_periods(sig, num=4, thresh=0.2, min_length=4, max_length=200)
    host_out, host_res = QOPeriods().find_periods(sig, num=4, thresh=0.2, min_length=4, max_length=200,
                                                  test_function=lambda self, x, y: rms(y) > rms(x) * 0.2)
    assert np.array_equal(dev_out["periods"], host_out["periods"]) and rel_err(host_out["norms"], dev_out["norms"]) < TOL
    assert dev_out["basis_dictionary"] == host_out["basis_dictionary"]
    assert rel_err(host_out["weights"], dev_out["weights"]) < 1e-8 and rel_err(host_res, dev_res) < 1e-8
    assert host_out["subspaces"].shape == dev_out["subspaces"].shape
    # update_weights=False (QOPeriods.py:645-714; overflows under numpy 2 in the v1 reference): every new period
    # is fitted to the running residual only, so the residual energy never grows and the weights concatenate
    out, res = QOPeriods().find_periods(sig, num=3, thresh=0.05, min_length=4, max_length=200, update_weights=False)
    assert len(out["periods"]) >= 2 and out["weights"].size == sum(out["basis_dictionary"].values())
    assert po.periodic_norm(res) < po.periodic_norm(sig)


def test_qoperiods_orthogonal_selection(eng):
    """find_periods(orthogonalize=True): raises TypeError in the v1 reference (best_base is never
    assigned, QOPeriods.py:427-448); offered as the commented-out lines intend.  Expectation restated
    from the oracle's pieces: period = argmax of the orthogonal powers of the residual, norm = gamma norm
    of its orthogonalised projection, weights re-solved over the natural-basis dictionary."""
    from pyperiod_amd import QOPeriods

    sig = multi_sinusoid_window(12, 900)
    got, res = QOPeriods(orthogonalize=True).find_periods(sig, num=3, thresh=0.05, max_length=200)
    periods, norms, r = [], [], sig.copy()
    for _ in range(3):
        p = po.best_period_orthogonal(r, 200, True)
        periods.append(p)
        norms.append(po.periodic_norm(po.project(r, p, False, True), p))
        a, dims = po.qo_get_subspaces(periods, sig.size)
        w, rec = po.qo_solve_quadratic(sig, a)
        r = sig - rec
        rms = lambda v: np.sqrt(np.mean(v * v))  # noqa: E731
        if not rms(rec) > rms(sig) * 0.05:
            break
    k = len(got["periods"])
    assert k >= 2 and list(got["periods"]) == periods[:k]
    assert rel_err(got["norms"], norms[:k]) < 1e-9


def test_fp32_algorithms_track_the_fp64_oracle(eng):
    """fp32 windows through the whole algorithms (build-defined path, the reference has none):
    period lists must agree with the fp64 oracle on the fp32-rounded input for well-separated
    signals, powers within 1e-4."""
    n = 2048
    x32 = multi_sinusoid_batch(200, 4, n, dtype=np.float32)
    x64 = x32.astype(np.float64)
    per, pw, bs, st = eng.m_best(x32, 4)
    assert bs.dtype == np.float32 and not st.any()
    counts, sper, spw, sbs, sst = eng.small_to_large(x32, 0.1, cap=24)
    ram = eng.ramanujan_norms(x32, 2, 128)
    for w in range(4):
        rper, rpw, rbs = po.m_best(x64[w], 4)
        assert np.array_equal(per[w], rper), w
        assert rel_err(pw[w], rpw) < 1e-4 and rel_err(bs[w], rbs) < 1e-4
        lper, lpw, _ = po.small_to_large(x64[w], 0.1)
        assert list(sper[w, : counts[w]]) == lper and rel_err(spw[w, : counts[w]], lpw) < 1e-4
        assert rel_err(ram[w], po.ramanujan_norms_folded(x64[w], 2, 128)) < 1e-5


def test_degenerate_windows(eng):
    """Constant, ramp, spike and exactly periodic noise-free windows: no hangs, no faults,
    statuses / values consistent with the oracle where the oracle is defined."""
    from pyperiod_amd import _ffi

    n = 512
    t = np.arange(n, dtype=np.float64)
    x = np.stack([np.full(n, 3.0), t / n, np.eye(1, n, 100)[0], np.sin(2 * np.pi * t / 8), np.zeros(n)])
    sw = eng.sweep(x, 2, 170, _ffi.PH_SWEEP_NORM)
    for w in range(5):
        assert rel_err(sw[w], po.sweep_norms(x[w], 2, 170)) < 1e-10 or np.allclose(sw[w], 0)
    per, pw, bs, st = eng.m_best(x, 3)
    assert st[4] == _ffi.PH_ST_NO_PERIOD  # all-zero window: the reference raises
    assert st[0] == _ffi.PH_ST_NO_PERIOD  # constant window: everything is removed by the first projection
    assert per[0, 0] == 2  # all norms tie, the lowest period wins (Periods.py:512)
    rper, rpw, _ = po.m_best(x[1], 3)
    assert st[1] == 0 and np.array_equal(per[1], rper) and rel_err(pw[1], rpw) < 1e-9
    counts, sper, spw, _, sst = eng.small_to_large(x, 0.05, want_bases=False)
    for w in (1, 2):
        lper, lpw, _ = po.small_to_large(x[w], 0.05)
        assert list(sper[w, : counts[w]]) == lper
    assert counts[4] == 0  # NaN comparisons never accept (Periods.py:281)
    ram = eng.ramanujan_norms(x, 2, 64)
    assert np.isfinite(ram).all()
    assert ram[3, 8] > 0.99 * ram[3].sum()  # a pure period-8 sinusoid lives in the q = 8 subspace


def test_long_windows_spill_second_buffer_to_hbm(eng):
    """N = 16384 fp64: two window-sized LDS buffers do not fit (256 KiB), so the projection
    buffer of the flagged paths and of m_best step 2 lives in an HBM workspace."""
    from pyperiod_amd import _ffi

    n = 16384
    x = multi_sinusoid_batch(300, 2, n)
    for trunc, orth in ((False, True), (True, True)):
        out = eng.project_batch(x, [37, 1260, 4096], trunc, orth)
        for w in range(2):
            for k, p in enumerate([37, 1260, 4096]):
                assert np.array_equal(out[w, k], po.project(x[w], p, trunc, orth)), (trunc, orth, p)
    sw = eng.sweep(x[:1], 2, 90, _ffi.PH_SWEEP_NORM, True, True)[0]
    assert rel_err(sw, po.sweep_norms(x[0], 2, 90, trunc=True, orth=True)) < TOL
    per, pw, bs, st = eng.m_best(x, 3, 400)
    for w in range(2):
        rper, rpw, rbs = po.m_best(x[w], 3, 400)
        assert st[w] == 0 and np.array_equal(per[w], rper)
        assert rel_err(pw[w], rpw) < TOL and rel_err(bs[w], rbs) < TOL
    counts, sper, spw, sbs, _ = eng.small_to_large(x[:1], 0.05, 300, True, False)
    lper, lpw, lbs = po.small_to_large(x[0], 0.05, 300, True, False)
    assert list(sper[0, : counts[0]]) == lper and rel_err(spw[0, : counts[0]], lpw) < TOL
    assert eng.max_window(np.float64, True, True) >= n


def test_windows_longer_than_lds_stream_from_hbm(eng):
    """N = 32768 fp64 (256 KiB) exceeds the 160 KiB LDS: the window itself lives in a per-workgroup
    HBM workspace and the same fold code reads it through L2 (kernels' <T, false> instantiations)."""
    from pyperiod_amd import _ffi

    n = 32768
    assert n > eng.max_window()
    x = multi_sinusoid_batch(500, 2, n)
    plist = [1, 3, 64, 97, 1260, 4099, 20000]
    out = eng.project_batch(x, plist)
    for w in range(2):
        for k, p in enumerate(plist):
            ref = po.project(x[w], p)
            assert np.array_equal(out[w, k], ref) if p > 1 else rel_err(out[w, k], ref) < 1e-12, p
    out = eng.project_batch(x[:1], [36, 1260], True, True)
    for k, p in enumerate([36, 1260]):
        assert np.array_equal(out[0, k], po.project(x[0], p, True, True)), p
    hi = 700
    sw = eng.sweep(x, 2, hi, _ffi.PH_SWEEP_NORM)
    sg = eng.sweep(x, 2, hi, _ffi.PH_SWEEP_NORM_GAMMA)
    sm = eng.sweep(x[:1], 2, 200, _ffi.PH_SWEEP_MAXABS)
    so = eng.sweep(x[:1], 2, 60, _ffi.PH_SWEEP_NORM, False, True)
    for w in range(2):
        assert rel_err(sw[w], po.sweep_norms(x[w], 2, hi)) < TOL
        assert rel_err(sg[w], po.sweep_norms(x[w], 2, hi, gamma=True)) < TOL
    assert rel_err(sm[0], po.sweep_maxabs(x[0], 2, 200)) < TOL
    assert rel_err(so[0], po.sweep_norms(x[0], 2, 60, orth=True)) < TOL
    per, pw, bs, st = eng.m_best(x, 3, 500)
    for w in range(2):
        rper, rpw, rbs = po.m_best(x[w], 3, 500)
        assert st[w] == 0 and np.array_equal(per[w], rper)
        assert rel_err(pw[w], rpw) < TOL and rel_err(bs[w], rbs) < TOL
    counts, sper, spw, sbs, _ = eng.small_to_large(x[:1], 0.05, 300)
    lper, lpw, lbs = po.small_to_large(x[0], 0.05, 300)
    assert list(sper[0, : counts[0]]) == lper and rel_err(spw[0, : counts[0]], lpw) < TOL
    assert rel_err(sbs[0, : counts[0]], np.array(lbs)) < TOL
    bper, bnr, bbs, bst = eng.best_correlation(x[:1], 2, 300)
    rper, rnr, rbs = po.best_correlation(x[0], 2, 300)
    assert np.array_equal(bper[0], rper) and rel_err(bnr[0], rnr) < TOL and rel_err(bbs[0], rbs) < TOL
    ram = eng.ramanujan_norms(x[:1], 2, 128)
    assert rel_err(ram[0], po.ramanujan_norms_folded(x[0], 2, 128)) < 1e-9
    # the remaining entry points (missing 5): folds, orthogonal powers, best_frequency, QOPeriods
    from pyperiod_amd import QOPeriods

    a, dims = po.qo_get_subspaces([37, 64, 101], n)
    folds = eng.fold_sums(x, [37, 64, 101], [37, 63, 100])
    assert rel_err(folds, x @ a.T) < 1e-12
    pows, ac, e3 = eng.orth_powers(x[:1], 400, True, want_autocorr=True, want_eq3=True)
    assert rel_err(pows[0], po.orth_powers(x[0], 400, True)) < TOL
    assert rel_err(ac[0, [0, 1, 7, 5000, n - 1]], [po.auto_corr(x[0], k) for k in (0, 1, 7, 5000, n - 1)]) < 1e-12
    fper, fpw, fbs, fst = eng.best_frequency(x[:1], None, 2)
    rper, rpw, rbs = po.best_frequency(x[0], None, 2)
    assert fst[0] == 0 and np.array_equal(fper[0], rper) and rel_err(fpw[0], rpw) < TOL and rel_err(fbs[0], rbs) < TOL
    fper, fpw, fbs, fst = eng.best_frequency(x[:1], None, 1, True, True)  # flagged projection of a long window
    rper, rpw, rbs = po.best_frequency(x[0], None, 1, True, True)
    assert np.array_equal(fper[0], rper) and rel_err(fbs[0], rbs) < TOL
    out, res = QOPeriods().find_periods(x[0], num=2, thresh=0.1, min_length=8, max_length=200)
    want, wres = po.qo_find_periods(x[0], 2, 0.1, 8, 200)
    assert np.array_equal(out["periods"], want["periods"]) and rel_err(out["norms"], want["norms"]) < TOL
    assert rel_err(out["weights"], want["weights"]) < 1e-8 and rel_err(res, wres) < 1e-8


def test_batch_interface_edges(eng):
    from pyperiod_amd import Periods, _ffi

    # one window, tiny window, odd lengths, largest window that fits
    for n in (2, 3, 17, 63, 64, 65, 4097):
        x = np.random.default_rng(n).standard_normal((2, n))
        hi = max(2, n // 2)
        sw = eng.sweep(x, 1, hi, _ffi.PH_SWEEP_NORM)
        for w in range(2):
            assert rel_err(sw[w], po.sweep_norms(x[w], 1, hi)) < TOL, n
    nmax = eng.max_window()
    x = multi_sinusoid_batch(0, 1, nmax)
    assert np.array_equal(eng.project_batch(x, [977])[0, 0], po.project(x[0], 977))
    with pytest.raises(ValueError):
        eng.project_batch(np.zeros((1, 8)), [0])
    with pytest.raises(ValueError):
        eng.sweep(np.zeros((1, 8)), 3, 2)
    with pytest.raises(ValueError):
        eng.sweep(np.zeros(8), 2, 3)  # not a batch
    # non-float64 input is upcast like numpy would
    xi = np.arange(24).reshape(2, 12)
    assert np.array_equal(eng.project_batch(xi, [5])[1, 0], po.project(xi[1].astype(float), 5))
    # batched class surface (extension over the reference)
    xb = multi_sinusoid_batch(0, 3, 512)
    per, pw, bs = Periods().m_best(xb, num=3)
    assert per.shape == (3, 3) and bs.shape == (3, 3, 512)
    res = Periods().small_to_large(xb, thresh=0.05)
    assert len(res) == 3 and res[1][0] == po.small_to_large(xb[1], 0.05)[0]


def test_config5_full_batch_fp32(eng, golden):
    """Config 5 at its per-GPU batch: 1024 fp32 windows of N = 16384 in ONE launch of ph_qo_find_periods
    (QOPeriods.py:313-596).  Every window finishes (status 0, no dictionary overflow), the batch equals the
    per-window calls bit for bit, two windows are checked against the fp64 oracle on the rounded input, and the two
    windows of the reference fixture `qoperiods_c5` sit inside the batch."""
    import torch

    n, W = 16384, 1024
    xb = multi_sinusoid_batch(0, W, n, dtype=np.float32)  # windows 0 and 7 are the fixture's
    xd = torch.from_numpy(xb).cuda()
    per, nrm, keeps, counts, wts, resid, st = [t.cpu().numpy() for t in eng.qo_find_periods(xd, 3, 0.1, 8, 300, 1024)]
    assert per.shape == (W, 3) and resid.shape == (W, n) and resid.dtype == np.float32
    assert int(np.abs(st).max()) == 0
    assert counts[:, 1].max() <= 3 and int(keeps.sum(axis=1).max()) <= 1024
    assert np.isfinite(resid).all() and np.isfinite(nrm).all()
    # the residual never carries more energy than the window
    assert (np.square(resid.astype(np.float64)).sum(axis=1) <= np.square(xb.astype(np.float64)).sum(axis=1) * (1 + 1e-6)).all()
    for w in (0, 511, 1023):  # batch == per-window
        one = eng.qo_find_periods(xb[w : w + 1], 3, 0.1, 8, 300, 1024)
        assert np.array_equal(one[0][0], per[w]) and np.array_equal(one[2][0], keeps[w]) and np.array_equal(one[3][0], counts[w])
        assert np.array_equal(one[1][0], nrm[w]) and np.array_equal(one[5][0], resid[w])
    for w in (3, 700):  # fp64 oracle on the rounded input
        out, res = po.qo_find_periods(xb[w].astype(np.float64), 3, 0.1, 8, 300)
        nrep, nb = counts[w]
        assert np.array_equal(per[w, :nrep], out["periods"]) and list(keeps[w, :nb]) == list(out["basis_dictionary"].values())
        k = int(keeps[w, :nb].sum())
        assert rel_err(nrm[w, :nrep], out["norms"]) < 1e-4 and rel_err(wts[w, :k], out["weights"]) < 1e-4
        assert rel_err(resid[w], res) < 1e-4
    g = golden("qoperiods_c5")  # the reference itself on window 0 (num=3, thresh 0.1)
    nrep, nb = counts[0]
    assert np.array_equal(per[0, :nrep], g["fp_w0_periods"]) and list(keeps[0, :nb]) == list(g["fp_w0_dict_vals"])
    assert rel_err(nrm[0, :nrep], g["fp_w0_norms"]) < 1e-4 and rel_err(resid[0], g["fp_w0_residual"]) < 1e-4
