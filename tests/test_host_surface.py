"""Host-side behaviour of the class surface that needs no GPU: argument conventions the reference has
(Periods.py) and that the drop-in keeps."""

import numpy as np
import pytest

from pyperiod_amd import Periods


def test_num_zero_returns_the_references_empty_arrays():
    """With num = 0 none of the reference's loops runs (Periods.py:316-320,376-389,488-501): it returns
    zero-length uint32 / float64 arrays and a (0, N) basis matrix.  No device work is needed."""
    x = np.sin(np.arange(240.0))
    for call in (lambda: Periods().m_best(x, num=0), lambda: Periods().m_best_gamma(x, num=0),
                 lambda: Periods().best_correlation(x, num=0), lambda: Periods().best_frequency(x, num=0)):
        per, pw, bs = call()
        assert per.shape == (0,) and per.dtype == np.uint32
        assert pw.shape == (0,) and pw.dtype == np.float64
        assert bs.shape == (0, 240)
    per, pw, bs = Periods().m_best(np.stack([x, x]), num=0)  # batch extension
    assert per.shape == (2, 0) and bs.shape == (2, 0, 240)


def test_non_array_and_2d_inputs_fail_like_the_reference():
    with pytest.raises(AttributeError):  # a list has no .size / .copy (Periods.py:171)
        Periods.project([1.0, 2.0, 3.0], 2)
    with pytest.raises(ValueError):  # 2-D input cannot be folded (Periods.py:176)
        Periods.project(np.zeros((2, 8)), 2)


def test_properties_keep_the_reference_quirks():
    p = Periods(True, False)
    assert p.trunc_to_integer_multiple == (True, False)  # the getter returns both flags (Periods.py:610-611)
    assert p.orthogonalize is False
    with pytest.raises(AttributeError):
        p.window  # noqa: B018 -- never set by __init__ (Periods.py:138-140)
