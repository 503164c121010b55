"""Drop-in ``RamanujanPeriods`` on the MI355X engine (reference pyPeriod/RamanujanPeriods.py).

``find_periods`` -- the Cq-dictionary correlation sweep of RamanujanPeriods.py:67-86 -- runs
as one kernel per window batch (ph_ramanujan_norms).  The kernel never builds the q x N
dictionary: it folds the window to period q and applies the exact integer Ramanujan-sum
filter, in float64.  The reference accumulates in float32 (RamanujanPeriods.py:127), so
parity with it is 1e-5 relative, not 1e-10.
"""

from __future__ import annotations

import numpy as np

from ._factors import get_factors, phi  # noqa: F401
from .engine import default_engine
from .Periods import _as_window, rms  # noqa: F401
from .QOPeriods import QOPeriods, flatten, ramanujan_sum  # noqa: F401


class RamanujanPeriods(QOPeriods):
    def __init__(self, basis_type="natural"):
        # the reference sets only three attributes (RamanujanPeriods.py:62-65) and therefore
        # fails later on the missing `_k`; the full QOPeriods attribute set is created here.
        super().__init__(basis_type)
        self._verbose = None

    def find_periods(self, x, min_length=2, max_length=None, select_periods=None):
        """Energy of the Ramanujan-subspace projection for every period (RamanujanPeriods.py:67-86).
        Returns ``norms`` of length max_length+1 (entries below min_length are 0)."""
        arr = np.asarray(x)
        batched = arr.ndim == 2
        win = np.ascontiguousarray(arr, dtype=np.float64) if batched else _as_window(x)[None, :]
        if not max_length:
            max_length = win.shape[1] // 3
        norms = default_engine().ramanujan_norms(win, int(min_length), int(max_length))
        norms = norms if batched else norms[0]
        if select_periods:
            if hasattr(select_periods, "__call__"):
                return select_periods(norms)
            return None  # the reference falls off the end here (RamanujanPeriods.py:82-84)
        return norms

    def find_periods_with_weights(self, x, min_length=2, max_length=None, thresh=0.2, **kwargs):
        """RamanujanPeriods.py:88-122 with the two v1 defects repaired (missing `_k`, and the
        swapped ``(weights, reconstruction)`` unpacking at :109-112)."""
        x = _as_window(x)
        norms = self.find_periods(x, min_length, max_length, select_periods=None)
        if "test_function" in kwargs:
            test_function = kwargs["test_function"]
        else:
            test_function = lambda v: np.argwhere(v / np.abs(np.max(v)) > thresh).flatten()  # noqa: E731
        periods = test_function(norms)
        basis_matricies, basis_dictionary = self.get_subspaces(periods, len(x))
        output_weights, reconstruction = self._solve_structured(x, basis_matricies, basis_dictionary)
        output_bases = {
            "periods": periods,
            "norms": norms[periods],
            "subspaces": basis_matricies,
            "weights": output_weights,
            "basis_dictionary": basis_dictionary,
        }
        self._output = output_bases
        return (output_bases, x - reconstruction)

    @staticmethod
    def project(x, basis):
        """row <- row / max(row); proj[i] = dot(x, row) * row, stored float32
        (RamanujanPeriods.py:124-131).  Runs on the GPU for any dictionary."""
        return default_engine().dict_project(np.asarray(x, dtype=np.float64), np.asarray(basis, dtype=np.float64))

    @staticmethod
    def Cq(q, s=0, repetitions=1, type="real"):
        """RamanujanPeriods.py:133-154."""
        return QOPeriods.Cq(q, s, repetitions, "complex" if type == "complex" else "real")

    def Cq_complete(self, q, N=None, normalize=True):
        """q circular shifts of c_q tiled to N, each L2-normalised (RamanujanPeriods.py:156-169)."""
        if N is None:
            N = q
        N = int(N)
        cq = self.Cq(q)
        reps = int(np.ceil(N / q))
        matrix = np.zeros((q, N))
        for i in range(q):
            matrix[i] = np.tile(np.roll(cq, i), reps)[:N]
            if normalize:
                matrix[i] /= np.linalg.norm(matrix[i])
        return matrix
