"""Exact integer helpers of the path (host side, tiny): divisor sets, primes, Euler phi, and
the set-order tables the device kernels consume.

Reference: get_factors Periods.py:55-84 (as defined) and RamanujanPeriods.py:25-39 (as
*called* at Periods.py:209,548), get_primes Periods.py:33-52, phi QOPeriods.py:16-43.

The reference iterates CPython ``set`` objects of divisors in two places (orthogonalisation,
Periods.py:209-214; m_best step 2, Periods.py:548-572) and its results depend on that order.
The tables below are built from real ``set`` objects filled in the same insertion order as the
reference, so the order is inherited from the interpreter instead of being re-modelled.
"""

from __future__ import annotations

import functools
import math

import numpy as np


def get_primes(max: int = 1000000) -> np.ndarray:
    """Primes <= max, ascending (Periods.py:33-52)."""
    flags = np.ones(max + 1, dtype=bool)
    flags[:2] = False
    for k in range(2, math.isqrt(max) + 1):
        if flags[k]:
            flags[k * k :: k] = False
    return np.flatnonzero(flags)


def get_factors(n, remove_1: bool = False, remove_n: bool = False, remove_1_and_n: bool = False) -> set:
    """Divisor set of n.  Accepts both reference spellings: (remove_1, remove_n) of
    Periods.py:55 / QOPeriods.py:46 and remove_1_and_n of RamanujanPeriods.py:25."""
    n = int(n)
    ordered = []
    for i in range(1, int(n ** 0.5) + 1):
        if n % i == 0:
            ordered += [i, n // i]
    facs = set(ordered)
    if remove_1 or remove_1_and_n:
        facs.remove(1)
    if remove_1_and_n:
        facs.remove(n)  # KeyError for n == 1, like RamanujanPeriods.py:36-38
    elif remove_n and n != 1:  # QOPeriods.py:73
        facs.remove(n)
    return facs


def phi(n: int) -> int:
    """Euler totient (QOPeriods.py:16-43)."""
    n = int(n)
    return sum(1 for k in range(1, n + 1) if math.gcd(n, k) == 1)


PRIMES = set(int(v) for v in get_primes(10000))  # Periods.py:121


def _csr(lists):
    off = np.zeros(len(lists) + 1, dtype=np.int32)
    np.cumsum([len(v) for v in lists], out=off[1:])
    flat = np.fromiter((v for lst in lists for v in lst), dtype=np.int32, count=int(off[-1]))
    if flat.size == 0:
        flat = np.zeros(1, dtype=np.int32)
    return off, flat


@functools.lru_cache(maxsize=8)
def factor_tables(max_p: int):
    """Dense CSR by period p in [0, max_p]: proper divisors of p (1 and p removed) in the
    iteration order of ``get_factors(p, remove_1_and_n=True)`` (Periods.py:548-549)."""
    lists = [[], []] + [list(get_factors(p, remove_1_and_n=True)) for p in range(2, max_p + 1)]
    return _csr(lists[: max_p + 1])


@functools.lru_cache(maxsize=8)
def orth_tables(max_p: int):
    """Dense CSR by period p: the sub-periods p // f, for every *prime* proper divisor f in
    set-iteration order, that Periods.project(orthogonalize=True) projects out
    (Periods.py:209-214)."""
    lists = [[], []]
    for p in range(2, max_p + 1):
        lists.append([p // f for f in get_factors(p, remove_1_and_n=True) if f in PRIMES])
    return _csr(lists[: max_p + 1])
