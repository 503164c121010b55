"""Seeded synthetic windows shared by the tests, the golden-vector
script and bench.py.  Pure numpy; no device code.

    x[w, n] = sum_{k<3} a_k sin(2 pi n / T_k + phi_k) + sigma * g[n]

with integer periods T_k drawn without replacement from [8, N/8], a_k ~ U(0.3, 1),
phi_k ~ U(0, 2 pi), g ~ N(0, 1), sigma = 0.05 (the noise breaks the exact norm ties that
noise-free integer-period signals have between p and 2p), generator default_rng(1000 + w).
"""

from __future__ import annotations

import numpy as np


def multi_sinusoid_window(w: int, n: int, sigma: float = 0.05, dtype=np.float64) -> np.ndarray:
    rng = np.random.default_rng(1000 + int(w))
    hi = max(10, n // 8)
    periods = rng.choice(np.arange(8, hi + 1), size=3, replace=False)
    amps = rng.uniform(0.3, 1.0, size=3)
    phases = rng.uniform(0.0, 2.0 * np.pi, size=3)
    noise = rng.standard_normal(n)
    t = np.arange(n, dtype=np.float64)
    x = sigma * noise
    for tk, ak, pk in zip(periods, amps, phases):
        x = x + ak * np.sin(2.0 * np.pi * t / float(tk) + pk)
    return x.astype(dtype)


def multi_sinusoid_batch(w0: int, count: int, n: int, sigma: float = 0.05, dtype=np.float64) -> np.ndarray:
    """Windows w0 .. w0+count-1 stacked row-major as (count, n)."""
    out = np.empty((count, n), dtype=dtype)
    for i in range(count):
        out[i] = multi_sinusoid_window(w0 + i, n, sigma, dtype)
    return out


def readme_window(n: int = 2000, seed: int = 0) -> np.ndarray:
    """The README-shaped two-sinusoid signal of config 1 (reference README.md:54-65) with
    the unseeded ``random.uniform`` noise replaced by default_rng(seed)."""
    sr, f1, f2, noise = 1000, 10, 17, 0.2
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    a = np.sin((t * np.pi * 2 * f1) / sr) + rng.uniform(-noise, noise, n)
    b = np.sin(((t * np.pi * 2 * f2) / sr) + (np.pi * 1.1)) + rng.uniform(-noise, noise, n)
    c = a + b
    return c / np.max(np.abs(c))
