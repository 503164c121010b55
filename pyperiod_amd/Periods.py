"""Drop-in ``Periods`` (Sethares & Staley periodicity transforms) on the MI355X engine.

Same constructor, method names, keyword arguments and return types as the reference class
(pyPeriod/Periods.py:90-644); every projection, norm and sweep runs in libperiod_hip.so.
Extension over the reference: each algorithm also accepts a ``(W, N)`` batch of independent
windows and then returns arrays/lists with a leading window axis.
"""

from __future__ import annotations

import math
from warnings import warn

import numpy as np

from . import _ffi
from ._factors import PRIMES, get_factors, get_primes  # noqa: F401  (re-exported like the reference module)
from .engine import default_engine


def rms(x) -> float:
    """Root mean square (Periods.py:16-30)."""
    return np.sqrt(np.sum(np.power(x, 2)) / len(x))


def _as_window(data):
    """The reference requires a 1-D ndarray: a list fails at ``data.copy()`` (AttributeError,
    Periods.py:171) and a 2-D array at the reshape (ValueError, Periods.py:176)."""
    data.size  # noqa: B018  -- AttributeError for non-arrays, like the reference
    arr = np.asarray(data)
    if arr.ndim != 1:
        raise ValueError(f"cannot fold an array of shape {arr.shape} into periods; expected 1-D")
    return np.ascontiguousarray(arr, dtype=np.float64)


def _raise_status(status, what):
    status = np.asarray(status)
    if np.any(status == _ffi.PH_ST_NO_PERIOD):
        # reference: max_base stays None -> TypeError at Periods.py:520/537 (project(data, None)
        # at Periods.py:334 for best_correlation)
        raise TypeError(f"{what}: no candidate period has a positive norm (all-zero or NaN window)")
    if np.any(status == _ffi.PH_ST_ITER_CAP):
        raise RuntimeError(f"{what}: iteration bound reached before `num` periods were found")


def _empty_result(x, num, batched):
    """num == 0: the reference's loops do not run and it returns its zero-length arrays
    (np.zeros(0, uint32), np.zeros(0), np.zeros((0, N)); Periods.py:316-318,376-378,488-490)."""
    w, n = x.shape
    per, pw, bs = np.zeros((w, 0), np.uint32), np.zeros((w, 0)), np.zeros((w, 0, n))
    return (per, pw, bs) if batched else (per[0], pw[0], bs[0])


class Periods:
    PRIMES = PRIMES  # Periods.py:121

    def __init__(self, trunc_to_integer_multiple: bool = False, orthogonalize: bool = False):
        self._trunc_to_integer_multiple = trunc_to_integer_multiple
        self._orthogonalize = orthogonalize

    # ------------------------------------------------------------------ leaf operations
    @staticmethod
    def project(data, p=2, trunc_to_integer_multiple=False, orthogonalize=False, return_single_period=False):
        """Projection onto the p-periodic subspace (Periods.py:142-219)."""
        x = _as_window(data)
        p = int(p)
        if p > x.size:
            warn("invalid value encountered in divide", RuntimeWarning)  # 0/0 columns, Periods.py:194
        if trunc_to_integer_multiple and getattr(data, "dtype", None) == np.float32:
            # np.mean on the float32 rectangle keeps float32 (Periods.py:178-184): row-order float32 sums and one
            # division -- what the library's float kernels do; nothing is computed in fp64 and cast back
            x = np.ascontiguousarray(data, dtype=np.float32)
        out = default_engine().project_batch(
            x[None, :], [p], bool(trunc_to_integer_multiple), bool(orthogonalize)
        )[0, 0]
        return out[0:p] if return_single_period else out

    @staticmethod
    def periodic_norm(x, p=None):
        """||x|| / sqrt(len(x)) [/ sqrt(p)]  (Periods.py:221-241)."""
        arr = np.ascontiguousarray(np.asarray(x), dtype=np.float64).reshape(1, -1)
        return np.float64(default_engine().periodic_norm(arr, p if p else None)[0])

    # ------------------------------------------------------------------ algorithms
    def _batch(self, data):
        arr = np.asarray(data)
        if arr.ndim == 2:
            return np.ascontiguousarray(arr, dtype=np.float64), True
        return _as_window(data)[None, :], False

    def small_to_large(self, data, thresh: float = 0.1, n_periods: int = None):
        """Small-to-large (Periods.py:246-287): three lists (periods, powers, bases)."""
        x, batched = self._batch(data)
        if n_periods is None:
            n_periods = math.floor(x.shape[1] / 2)
        counts, per, pw, bs, _ = default_engine().small_to_large(
            x, thresh, n_periods, self._trunc_to_integer_multiple, self._orthogonalize
        )
        res = []
        for w in range(x.shape[0]):
            k = int(counts[w])
            res.append(
                ([int(v) for v in per[w, :k]], [np.float64(v) for v in pw[w, :k]], [bs[w, i].copy() for i in range(k)])
            )
        return res if batched else res[0]

    def best_correlation(self, data, num: int = 5, max_length: int = None, ratio: float = 0.01):
        """Best-correlation (Periods.py:289-349)."""
        x, batched = self._batch(data)
        if max_length is None:
            max_length = math.floor(x.shape[1] / 3)
        if num == 0:
            return _empty_result(x, num, batched)
        per, nr, bs, st = default_engine().best_correlation(
            x, num, max_length, ratio, self._trunc_to_integer_multiple, self._orthogonalize
        )
        _raise_status(st, "best_correlation")
        return (per, nr, bs) if batched else (per[0], nr[0], bs[0])

    def best_frequency(self, data, win_size: int = None, num: int = 5):
        """Best-frequency (Periods.py:351-398) on the GPU: spectral peak of the rfft of length win_size (in-LDS
        radix-2 FFT, Bluestein for other lengths, direct DFT beyond the LDS), p = round(2 win_size / k),
        project, subtract -- one call for all `num` rounds."""
        x, batched = self._batch(data)
        n = x.shape[1]
        if win_size is None:
            win_size = n
        elif win_size < n:
            warn("win_size is smaller than the input signal length. It will be truncated and information will be lost.")
        if num == 0:
            return _empty_result(x, num, batched)
        per, pw, bs, st = default_engine().best_frequency(
            x, win_size, num, self._trunc_to_integer_multiple, self._orthogonalize
        )
        if np.any(np.asarray(st) != 0):
            # the spectral peak was bin 0: the reference evaluates 2 * win_size / 0 and int(round(inf))
            raise OverflowError("cannot convert float infinity to integer")
        return (per, pw, bs) if batched else (per[0], pw[0], bs[0])

    def m_best(self, data, num: int = 5, max_length: int = None, min_length: int = 2):
        """M-best (Periods.py:408-430)."""
        return self._m_best_meta(data, None, num, max_length, min_length)

    def m_best_gamma(self, data, num: int = 5, max_length: int = None, min_length: int = 2):
        """M-best gamma (Periods.py:432-454)."""
        return self._m_best_meta(data, "gamma", num, max_length, min_length)

    def _m_best_meta(self, data, type, num=5, max_length=None, min_length=2):
        """Periods.py:456-601: both steps run on the device (one launch pair per batch)."""
        if self.orthogonalize:
            warn("`Orthogonalize = True` has no effect in M-best.")  # Periods.py:482-483
        x, batched = self._batch(data)
        if max_length is None:
            max_length = math.floor(x.shape[1] / 3)
        if num == 0:
            return _empty_result(x, num, batched)
        per, pw, bs, st = default_engine().m_best(
            x, num, max_length, min_length, type is not None, self._trunc_to_integer_multiple, self._orthogonalize
        )
        _raise_status(st, "m_best")
        return (per, pw, bs) if batched else (per[0], pw[0], bs[0])

    # ------------------------------------------------------------------ properties (Periods.py:606-644)
    @property
    def trunc_to_integer_multiple(self):
        # the reference getter returns BOTH flags as a tuple (Periods.py:610-611)
        return self._trunc_to_integer_multiple, self._orthogonalize

    @trunc_to_integer_multiple.setter
    def trunc_to_integer_multiple(self, value):
        self._trunc_to_integer_multiple, self._orthogonalize = value

    @property
    def orthogonalize(self):
        return self._orthogonalize

    @orthogonalize.setter
    def orthogonalize(self, value):
        self._orthogonalize = value

    @property
    def window(self):
        return self._window  # never set by __init__, like the reference (Periods.py:138-140)

    @window.setter
    def window(self, value):
        self._window = value
