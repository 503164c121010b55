#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the *reference* (pyPeriod v1,
mounted read-only at /root/reference) on seeded inputs.  Build container only: the
reference never travels to the GPU box, the .npz files written here do.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [section ...]

Sections (default: all): kat project project_f32 sweep small_to_large m_best best_correlation ramanujan
qoperiods orth_powers m_best_split ramanujan_c3 ramanujan_default ramanujan_weights
qoperiods_c5 m_best_large_p.

Shims -- none of them changes reference arithmetic:
  1. ``builtins.Any = typing.Any`` so that ``import pyPeriod`` survives QOPeriods.py:86.
  2. the reference's own ``get_factors(n, remove_1_and_n)`` (RamanujanPeriods.py:25-39) is
     bound into the Periods module namespace, because Periods.py:209,548 call it with that
     keyword while Periods.py:55 defines a different signature.
  3. QOPeriods cannot be constructed (QOPeriods.py:190); a ``class QO(QOPeriods, Periods)``
     instance is created with ``object.__new__`` and the attribute list of QOPeriods.py:191-199.
  4. RamanujanPeriods.find_periods_with_weights (RamanujanPeriods.py:88-122) dies on two v1
     defects: ``__init__`` (:62-65) never sets ``_k`` although get_subspaces reads it
     (QOPeriods.py:844), and :109 unpacks ``solve_quadratic``'s ``(weights, reconstruction)``
     as ``(reconstruction, weights)``.  The harness sets ``_k = 0`` on the instance and binds an
     instance-level ``solve_quadratic`` that calls the reference's own static method and hands
     the pair back in the order :109 expects.  The arithmetic is the reference's.
  (m_best_split only) ``numpy.insert`` is wrapped with a call counter inside the reference's
     Periods module so that the number of step-2 splits (Periods.py:581-594) is recorded.
Only data (inputs + the reference's outputs) is stored; no reference source.
"""

import builtins
import contextlib
import io
import os
import sys
import typing
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from pyperiod_amd.synth import multi_sinusoid_window, readme_window  # noqa: E402


def load_reference():
    builtins.Any = typing.Any  # shim 1
    sys.path.insert(0, "/root/reference")
    import pyPeriod  # noqa: F401

    per_mod = sys.modules["pyPeriod.Periods"]
    ram_mod = sys.modules["pyPeriod.RamanujanPeriods"]
    qo_mod = sys.modules["pyPeriod.QOPeriods"]
    per_mod.get_factors = ram_mod.get_factors  # shim 2
    return per_mod, ram_mod, qo_mod


def make_qo(QOP, Periods):
    class QO(QOP, Periods):  # shim 3
        pass

    qo = object.__new__(QO)
    qo._trunc_to_integer_multiple = False
    qo._orthogonalize = False
    qo._output = None
    qo._basis_type = "natural"
    qo._verbose = False
    qo._k = 0
    qo._window = False
    qo._output_bases = None
    qo._container = []
    return qo


def main():
    warnings.simplefilter("ignore")
    per_mod, ram_mod, qo_mod = load_reference()
    Periods = per_mod.Periods
    Ram = ram_mod.RamanujanPeriods
    FLAGS = [(False, False), (True, False), (False, True), (True, True)]
    c1 = readme_window(2000, 0)
    QOP = qo_mod.QOPeriods
    qo = make_qo(QOP, Periods)
    asked = [a for a in sys.argv[1:] if not a.startswith("-")]

    def want(section):
        return not asked or section in asked

    # ---------------------------------------------------------------- KATs
    if want("kat"):
        kat = {
            "project_arange10_p3": Periods.project(np.arange(10.0), 3),
            "project_arange10_p3_trunc": Periods.project(np.arange(10.0), 3, True),
            "norm_arange10": np.float64(Periods.periodic_norm(np.arange(10.0))),
            "norm_arange10_p3": np.float64(Periods.periodic_norm(np.arange(10.0), 3)),
            "cq6": Ram.Cq(6),
            "phi_9_10": np.array([qo_mod.phi(9), qo_mod.phi(10)]),
            "n_primes_10000": np.int64(len(Periods.PRIMES)),
        }
        # divisor-set iteration order as the reference sees it (CPython set order)
        order_n = np.arange(2, 1400)
        kat["factor_order_n"] = order_n
        flat, off = [], [0]
        for n in order_n:
            flat += [int(v) for v in per_mod.get_factors(int(n), remove_1_and_n=True)]
            off.append(len(flat))
        kat["factor_order_flat"] = np.array(flat, dtype=np.int64)
        kat["factor_order_off"] = np.array(off, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, "kat.npz"), **kat)

    # ---------------------------------------------------------------- project
    if want("project"):
        out = {}
        for n in (10, 97, 240, 4096):
            x = multi_sinusoid_window(7, n) if n >= 64 else np.random.default_rng(n).standard_normal(n)
            out[f"x_{n}"] = x
            for p in (2, 3, 7, 12, 64, 97, n // 2):
                if p > n or p < 2:
                    continue
                for trunc, orth in FLAGS:
                    full = Periods.project(x, p, trunc, orth)
                    single = Periods.project(x, p, trunc, orth, True)
                    key = f"n{n}_p{p}_t{int(trunc)}_o{int(orth)}"
                    assert np.array_equal(full[:p], single)
                    if n <= 240:
                        out[key] = full
                    else:
                        # every projection is p-periodic; the first period determines it
                        assert np.array_equal(np.tile(single, n // p + 1)[:n], full)
                        out[key + "_single"] = single
        np.savez_compressed(os.path.join(HERE, "project.npz"), **out)

    # ---------------------------------------------------------------- project, float32 windows in trunc mode
    # (np.mean on the float32 rectangle keeps float32, Periods.py:178-184: row-order float32 sums, one division)
    if want("project_f32"):
        out = {}
        for n in (97, 240, 4096):
            x = multi_sinusoid_window(9, n, dtype=np.float32)
            assert x.dtype == np.float32
            out[f"x_{n}"] = x
            for p in (2, 3, 7, 12, 64, 97, n // 2):
                if p > n or p < 2:
                    continue
                for orth in (False, True):
                    full = Periods.project(x, p, True, orth)
                    assert full.dtype == np.float32
                    out[f"n{n}_p{p}_o{int(orth)}"] = full
        np.savez_compressed(os.path.join(HERE, "project_f32.npz"), **out)

    # ---------------------------------------------------------------- sweeps (N=4096, 4 windows)
    if want("sweep"):
        out = {}
        n = 4096
        p_lo, p_hi = 2, n // 3
        for w in range(4):
            x = multi_sinusoid_window(w, n)
            plain = np.zeros(p_hi - p_lo + 1)
            gamma = np.zeros_like(plain)
            maxabs = np.zeros_like(plain)
            for k, p in enumerate(range(p_lo, p_hi + 1)):
                base = Periods.project(x, p, False, False)
                plain[k] = Periods.periodic_norm(base)
                gamma[k] = Periods.periodic_norm(base, p)
                maxabs[k] = max(abs(sum(x[s::p])) for s in range(p)) if w == 0 else np.nan
            out[f"plain_w{w}"] = plain
            out[f"gamma_w{w}"] = gamma
            if w == 0:
                out["maxabs_w0"] = maxabs
        x = multi_sinusoid_window(0, n)
        for trunc, orth in FLAGS[1:]:
            out[f"plain_w0_t{int(trunc)}_o{int(orth)}"] = np.array(
                [Periods.periodic_norm(Periods.project(x, p, trunc, orth)) for p in range(p_lo, p_hi + 1)]
            )
        np.savez_compressed(os.path.join(HERE, "sweep.npz"), **out)

    # ---------------------------------------------------------------- small_to_large
    if want("small_to_large"):
        out = {}
        per, pw, bs = Periods().small_to_large(c1, thresh=0.1)  # BASELINE config 1
        out["c1_periods"], out["c1_powers"], out["c1_bases"] = np.array(per), np.array(pw), np.array(bs)
        for w in range(4):
            x = multi_sinusoid_window(w, 4096)
            per, pw, bs = Periods().small_to_large(x, thresh=0.05)  # config 4 unit
            out[f"w{w}_periods"], out[f"w{w}_powers"] = np.array(per), np.array(pw)
            if w == 1:
                out["w1_bases"] = np.array(bs)
        for trunc, orth in FLAGS[1:]:
            x = multi_sinusoid_window(2, 1200)
            per, pw, bs = Periods(trunc, orth).small_to_large(x, thresh=0.05)
            tag = f"n1200_t{int(trunc)}_o{int(orth)}"
            out[tag + "_periods"], out[tag + "_powers"], out[tag + "_bases"] = (
                np.array(per),
                np.array(pw),
                np.array(bs).reshape(len(per), 1200),
            )
        x = multi_sinusoid_window(3, 600)
        per, pw, bs = Periods().small_to_large(x, thresh=0.02, n_periods=100)
        out["n600_np100_periods"], out["n600_np100_powers"] = np.array(per), np.array(pw)
        np.savez_compressed(os.path.join(HERE, "small_to_large.npz"), **out)

    # ---------------------------------------------------------------- m_best / m_best_gamma
    if want("m_best"):
        out = {}
        for name in ("m_best", "m_best_gamma"):
            for w in range(4):
                x = multi_sinusoid_window(w, 4096)
                per, pw, bs = getattr(Periods(), name)(x, num=10)  # config 2 unit
                out[f"{name}_w{w}_periods"], out[f"{name}_w{w}_powers"] = per, pw
                if w == 1:
                    out[f"{name}_w1_bases"] = bs
            for w in (4, 5):
                x = multi_sinusoid_window(w, 1500)
                per, pw, bs = getattr(Periods(), name)(x, num=6, max_length=300, min_length=3)
                out[f"{name}_n1500_w{w}_periods"], out[f"{name}_n1500_w{w}_powers"] = per, pw
                out[f"{name}_n1500_w{w}_bases"] = bs
            # README signal (config 1 shape)
            per, pw, bs = getattr(Periods(), name)(c1, num=10)
            out[f"{name}_c1_periods"], out[f"{name}_c1_powers"], out[f"{name}_c1_bases"] = per, pw, bs
            # flag variants (orthogonalize *does* reach project, Periods.py:504-506)
            for trunc, orth in FLAGS[1:]:
                x = multi_sinusoid_window(6, 900)
                per, pw, bs = getattr(Periods(trunc, orth), name)(x, num=5)
                tag = f"{name}_n900_t{int(trunc)}_o{int(orth)}"
                out[tag + "_periods"], out[tag + "_powers"], out[tag + "_bases"] = per, pw, bs
        np.savez_compressed(os.path.join(HERE, "m_best.npz"), **out)

    # ---------------------------------------------------------------- best_correlation / best_frequency
    if want("best_correlation"):
        out = {}
        x = multi_sinusoid_window(1, 4096)
        per, nr, bs = Periods().best_correlation(x, num=3)
        out["bc_n4096_periods"], out["bc_n4096_norms"], out["bc_n4096_bases"] = per, nr, bs
        for w in (2, 3):
            x = multi_sinusoid_window(w, 700)
            per, nr, bs = Periods().best_correlation(x, num=5, ratio=0.01)
            out[f"bc_n700_w{w}_periods"], out[f"bc_n700_w{w}_norms"], out[f"bc_n700_w{w}_bases"] = per, nr, bs
        per, pw, bs = Periods().best_frequency(c1, win_size=None, num=4)
        out["bf_c1_periods"], out["bf_c1_powers"], out["bf_c1_bases"] = per, pw, bs
        np.savez_compressed(os.path.join(HERE, "best_correlation.npz"), **out)

    # ---------------------------------------------------------------- Ramanujan
    if want("ramanujan"):
        out = {}
        ram = Ram()
        for q in range(1, 65):
            out[f"cq_{q}"] = Ram.Cq(q)
        out["cq_complete_6_20"] = ram.Cq_complete(6, 20)
        x = multi_sinusoid_window(0, 240)
        out["norms_n240"] = ram.find_periods(x, 2, 80)
        x = multi_sinusoid_window(1, 8192)
        out["norms_n8192_pmax64"] = ram.find_periods(x, 2, 64)
        x = multi_sinusoid_window(2, 1000)
        out["norms_n1000_default"] = ram.find_periods(x)  # max_length = N // 3
        np.savez_compressed(os.path.join(HERE, "ramanujan.npz"), **out)

    # ---------------------------------------------------------------- QOPeriods pieces (config 5)
    if want("qoperiods"):
        out = {}
        a_mat, dims = qo.get_subspaces([37, 64, 101], 16384)
        out["dims_37_64_101_keys"] = np.array([int(k) for k in dims.keys()])
        out["dims_37_64_101_vals"] = np.array([int(v) for v in dims.values()])
        a_mat, dims = qo.get_subspaces([12, 18, 8, 5], 1024)
        out["dims_12_18_8_5_vals"] = np.array([int(v) for v in dims.values()])
        x = multi_sinusoid_window(3, 1024)
        wts, rec = QOP.solve_quadratic(x, a_mat)
        out["solve_x"], out["solve_w"], out["solve_recon"] = x, wts, rec
        out["pp_5_12_keep3"] = QOP.Pp(5, 12, keep=3)
        for tag, sig, kw in (
            ("c1", c1, dict(num=2, thresh=0.05)),
            ("w5", multi_sinusoid_window(5, 1536), dict(num=4, thresh=0.2, min_length=4, max_length=200)),
        ):
            with contextlib.redirect_stdout(io.StringIO()):  # QOPeriods.py:488 prints unconditionally
                res_out, res = qo.find_periods(sig, **kw)
            out[f"fp_{tag}_periods"] = np.asarray(res_out["periods"])
            out[f"fp_{tag}_norms"] = np.asarray(res_out["norms"])
            out[f"fp_{tag}_weights"] = np.asarray(res_out["weights"])
            out[f"fp_{tag}_dict_keys"] = np.array([int(k) for k in res_out["basis_dictionary"].keys()])
            out[f"fp_{tag}_dict_vals"] = np.array([int(v) for v in res_out["basis_dictionary"].values()])
            out[f"fp_{tag}_residual"] = res
        np.savez_compressed(os.path.join(HERE, "qoperiods.npz"), **out)

    # ---------------------------------------------------------------- orthogonal period powers
    if want("orth_powers"):
        out = {}
        for tag, sig, max_p in (
            ("w1_n600", multi_sinusoid_window(1, 600), 200),
            ("w2_n1000", multi_sinusoid_window(2, 1000), None),
            ("c1", c1, 400),
        ):
            out[f"pows_{tag}"] = qo.get_best_period_orthogonal(sig, max_p, normalize=False, return_powers=True)
            out[f"pows_norm_{tag}"] = qo.get_best_period_orthogonal(sig, max_p, normalize=True, return_powers=True)
            out[f"best_{tag}"] = np.int64(qo.get_best_period_orthogonal(sig, max_p, normalize=True))
            out[f"best_raw_{tag}"] = np.int64(qo.get_best_period_orthogonal(sig, max_p))
        sig = multi_sinusoid_window(1, 600)
        out["eq3_w1_n600"] = np.array([qo.eq_3(sig, q) for q in range(1, 60)])
        out["autocorr_w1_n600"] = np.array([qo.auto_corr(sig, k) for k in range(0, 600, 7)])
        np.savez_compressed(os.path.join(HERE, "orth_powers.npz"), **out)

    # ---------------------------------------------------------------- m_best_gamma with real step-2 splits
    if want("m_best_split"):
        # (N, max_length, num, window): found by scanning seeds with the oracle for calls in which
        # step 2 of the gamma variant (stale-`p` divisor, Periods.py:559,572) accepts a split
        out = {}
        calls = [0]
        real_insert = np.insert

        def counting_insert(*a, **k):
            calls[0] += 1
            return real_insert(*a, **k)

        cases = [(240, 16, 4, 22), (240, 30, 5, 11), (600, 20, 5, 4), (600, 16, 3, 4), (240, 20, 5, 22)]
        try:
            per_mod.np.insert = counting_insert
            for name in ("m_best_gamma", "m_best"):
                for n, ml, num, w in cases:
                    x = multi_sinusoid_window(w, n)
                    calls[0] = 0
                    per, pw, bs = getattr(Periods(), name)(x, num=num, max_length=ml)
                    tag = f"{name}_n{n}_ml{ml}_num{num}_w{w}"
                    out[tag + "_periods"], out[tag + "_powers"], out[tag + "_bases"] = per, pw, bs
                    out[tag + "_splits"] = np.int64(calls[0] // 3)  # three np.insert calls per split (:585-594)
        finally:
            per_mod.np.insert = real_insert
        out["cases"] = np.array(cases, dtype=np.int64)
        assert all(int(out[f"m_best_gamma_n{n}_ml{ml}_num{num}_w{w}_splits"]) >= 1 for n, ml, num, w in cases)
        np.savez_compressed(os.path.join(HERE, "m_best_split.npz"), **out)

    # ---------------------------------------------------------------- Ramanujan at BASELINE config 3's shape
    if want("ramanujan_c3"):
        out = {}
        ram = Ram()
        for w in (0, 1):  # ~35 s per window: N = 8192, q = 2 .. 512
            out[f"norms_n8192_pmax512_w{w}"] = ram.find_periods(multi_sinusoid_window(w, 8192), 2, 512)
        np.savez_compressed(os.path.join(HERE, "ramanujan_c3.npz"), **out)

    # ---------------------------------------------------------------- Ramanujan, default range at N = 4096
    if want("ramanujan_default"):
        # max_length = N // 3 = 1365 (RamanujanPeriods.py:68-69); ~15 min: Cq is O(q phi(q)) interpreted
        out = {"norms_n4096_default_w3": Ram().find_periods(multi_sinusoid_window(3, 4096))}
        np.savez_compressed(os.path.join(HERE, "ramanujan_default.npz"), **out)

    # ---------------------------------------------------------------- find_periods_with_weights (shim 4)
    if want("ramanujan_weights"):
        out = {}
        ram = Ram()
        ram._k = 0  # shim 4a
        ram._window = False
        ram.solve_quadratic = lambda x, a: QOP.solve_quadratic(x, a)[::-1]  # shim 4b: order expected at :109
        for tag, sig, kw in (
            ("n240", multi_sinusoid_window(0, 240), dict(min_length=2, max_length=80, thresh=0.2)),
            ("n1000", multi_sinusoid_window(2, 1000), dict(thresh=0.3)),
            ("n600", multi_sinusoid_window(5, 600), dict(min_length=3, max_length=150, thresh=0.1)),
        ):
            res_out, res = ram.find_periods_with_weights(sig, **kw)
            out[f"{tag}_periods"] = np.asarray(res_out["periods"])
            out[f"{tag}_norms"] = np.asarray(res_out["norms"])
            out[f"{tag}_weights"] = np.asarray(res_out["weights"])
            out[f"{tag}_dict_keys"] = np.array([int(k) for k in res_out["basis_dictionary"].keys()])
            out[f"{tag}_dict_vals"] = np.array([int(v) for v in res_out["basis_dictionary"].values()])
            out[f"{tag}_residual"] = np.asarray(res)
        np.savez_compressed(os.path.join(HERE, "ramanujan_weights.npz"), **out)

    # ---------------------------------------------------------------- QOPeriods.find_periods at config 5's length
    if want("qoperiods_c5"):
        out = {}
        for tag, w, kw in (
            ("w0", 0, dict(num=3, thresh=0.1, min_length=8, max_length=300)),
            ("w7", 7, dict(num=4, thresh=0.05, min_length=8, max_length=300)),
        ):
            # the fp32 window of config 5, handed to the (fp64) reference after rounding
            sig = multi_sinusoid_window(w, 16384, dtype=np.float32).astype(np.float64)
            with contextlib.redirect_stdout(io.StringIO()):
                res_out, res = qo.find_periods(sig, **kw)
            out[f"fp_{tag}_periods"] = np.asarray(res_out["periods"])
            out[f"fp_{tag}_norms"] = np.asarray(res_out["norms"])
            out[f"fp_{tag}_weights"] = np.asarray(res_out["weights"])
            out[f"fp_{tag}_dict_keys"] = np.array([int(k) for k in res_out["basis_dictionary"].keys()])
            out[f"fp_{tag}_dict_vals"] = np.array([int(v) for v in res_out["basis_dictionary"].values()])
            out[f"fp_{tag}_residual"] = res.astype(np.float32)  # compared at 1e-4 (fp32 config)
        np.savez_compressed(os.path.join(HERE, "qoperiods_c5.npz"), **out)

    # ---------------------------------------------------------------- m_best with max_length far above N/3
    if want("m_best_large_p"):
        # periods beyond N/2 (single-sample residues) and beyond 2N/3; the first case splits 454 -> 227 in step 2.
        # The inputs are stored: they are not one of the synth generators.
        out = {}
        cases = [(600, 500, 4, 0), (257, 210, 3, 1), (1024, 900, 5, 2)]
        for n, ml, num, seed in cases:
            rng = np.random.default_rng(seed)
            x = rng.standard_normal(n) + 2.0 * np.sin(2 * np.pi * np.arange(n) / (0.8 * n))
            out[f"x_n{n}"] = x
            for name in ("m_best", "m_best_gamma"):
                per, pw, bs = getattr(Periods(), name)(x, num=num, max_length=ml)
                tag = f"{name}_n{n}_ml{ml}_num{num}"
                out[tag + "_periods"], out[tag + "_powers"], out[tag + "_bases"] = per, pw, bs
        out["cases"] = np.array(cases, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, "m_best_large_p.npz"), **out)

    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print(f"golden fixtures written to {HERE}: {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
