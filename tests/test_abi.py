"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/periodhip.h declares; the host layer fails loudly without a GPU."""

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "periodhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ph_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    from pyperiod_amd import _ffi

    return _ffi.load()


def test_every_declared_symbol_is_exported(lib):
    from pyperiod_amd import _ffi

    names = _header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in periodhip.h but not exported"
    assert sorted(list(_ffi.SIGNATURES) + ["ph_last_error", "ph_profile_name"]) == names
    assert lib.ph_version() == 100


def test_header_constants_match_binding():
    from pyperiod_amd import _ffi

    text = open(os.path.join(ROOT, "include", "periodhip.h")).read()
    for name in ("PH_OK", "PH_E_ARG", "PH_E_HIP", "PH_E_NOMEM", "PH_E_CAP", "PH_E_UNSUPPORTED", "PH_F64", "PH_F32",
                 "PH_FLAG_TRUNC", "PH_FLAG_ORTH", "PH_FLAG_SINGLE", "PH_FLAG_DEVICE", "PH_FLAG_NOSYNC", "PH_SWEEP_NORM",
                 "PH_SWEEP_NORM_GAMMA", "PH_SWEEP_MAXABS", "PH_ST_OK", "PH_ST_NO_PERIOD", "PH_ST_ITER_CAP", "PH_ST_CAP"):
        m = re.search(rf"#define {name} \(?(-?\d+)u?\)?", text)
        assert m, name
        assert int(m.group(1)) == getattr(_ffi, name), name


def test_argument_errors_without_gpu(lib):
    from pyperiod_amd import _ffi

    # NULL context -> PH_E_ARG with a message, no crash, no GPU needed
    rc = lib.ph_sweep(None, None, 0, 1, 16, 2, 5, 0, None, None, 0, 0, None)
    assert rc == _ffi.PH_E_ARG
    assert b"ctx" in lib.ph_last_error()
    with pytest.raises(ValueError):
        _ffi.check(rc)


def test_helper_entry_points_reject_null_without_gpu(lib):
    """The round-2 helpers (pass-plan info, QOPeriods feasibility) validate their arguments on the host."""
    from pyperiod_amd import _ffi

    n_pass, n_per, ok = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(7)
    assert lib.ph_sweep_plan_info(None, 2, 1365, ctypes.byref(n_pass), ctypes.byref(n_per)) == _ffi.PH_E_ARG
    assert lib.ph_qo_feasible(None, _ffi.PH_F64, 4096, -1, 512, ctypes.byref(ok)) == _ffi.PH_E_ARG
    assert lib.ph_small_to_large(None, None, 0, 1, 16, 0.1, -1, None, None, 0, _ffi.PH_FLAG_DEVICE | _ffi.PH_FLAG_NOSYNC, 4,
                                 None, None, None, None, None) == _ffi.PH_E_ARG
    assert _ffi.PH_FLAG_NOSYNC == 16


def test_product_has_no_cpu_fallback(lib):
    """Without a GPU the class surface must raise, not compute on the host."""
    n = ctypes.c_int(0)
    lib.ph_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("GPU present")
    from pyperiod_amd import Periods
    from pyperiod_amd._ffi import PeriodHipError

    with pytest.raises(PeriodHipError):
        Periods.project(np.arange(10.0), 3)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pyperiod_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "period_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_set_order_tables_match_reference(golden):
    """The CSR tables fed to the kernels reproduce the reference's divisor-set iteration order."""
    from pyperiod_amd import _factors

    kat = golden("kat")
    off = kat["factor_order_off"]
    foff, fq = _factors.factor_tables(1399)
    ooff, oq = _factors.orth_tables(1399)
    for k, n in enumerate(kat["factor_order_n"]):
        want = [int(v) for v in kat["factor_order_flat"][off[k] : off[k + 1]]]
        assert list(fq[foff[n] : foff[n + 1]]) == want, n
        assert list(oq[ooff[n] : ooff[n + 1]]) == [n // f for f in want if f in _factors.PRIMES], n
    assert _factors.phi(9) == 6 and _factors.phi(10) == 4
    assert len(_factors.PRIMES) == 1229


def test_host_dictionaries_match_reference(golden):
    """Integer Ramanujan sums / natural-basis rows built on the host (no GPU involved)."""
    from pyperiod_amd.QOPeriods import QOPeriods, ramanujan_sum

    g = golden("ramanujan")
    for q in range(1, 65):
        assert np.array_equal(ramanujan_sum(q), np.rint(g[f"cq_{q}"]).astype(np.int64))
    gq = golden("qoperiods")
    assert np.array_equal(QOPeriods.Pp(5, 12, keep=3), gq["pp_5_12_keep3"])
    qo = QOPeriods()
    a, d = qo.get_subspaces([37, 64, 101], 256)
    assert list(d.values()) == [37, 63, 100] and a.shape == (200, 256)
    assert [int(k) for k in d] == list(gq["dims_37_64_101_keys"])
    from pyperiod_amd import RamanujanPeriods

    assert np.max(np.abs(RamanujanPeriods().Cq_complete(6, 20) - g["cq_complete_6_20"])) < 1e-12
