"""Batched host planner over the C ABI: one `PeriodEngine` per GPU.

Every method takes a batch of independent signal windows, row-major ``(W, N)``:
  * a numpy array  -> host-pointer call; the library stages data through its own device
    buffers and returns numpy arrays (this is what the drop-in classes use);
  * a torch CUDA tensor -> device-pointer call on torch's current stream; outputs are torch
    tensors on the same device and nothing crosses PCIe (what bench.py and the sharded
    multi-GPU path use).

No arithmetic of the path happens here; this file only shapes arguments for
``libperiod_hip.so`` and raises if the library or the GPU is missing.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _factors, _ffi

_NP_DTYPES = {np.dtype(np.float64): _ffi.PH_F64, np.dtype(np.float32): _ffi.PH_F32}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class _Out:
    """Allocates outputs next to the input (numpy or torch) and hands out raw addresses."""

    def __init__(self, like):
        self.torch = _is_torch(like)
        self.stream = None  # hipStream_t handle the call must run on (None = the context's own)
        if self.torch:
            import torch

            self._t = torch
            self.device = like.device

    def empty(self, shape, dtype):
        if self.torch:
            tdt = {
                np.float64: self._t.float64,
                np.float32: self._t.float32,
                np.int32: self._t.int32,
                np.uint32: self._t.int32,  # same bits; viewed back by the caller if needed
            }[dtype]
            return self._t.empty(shape, dtype=tdt, device=self.device)
        return np.empty(shape, dtype=dtype)

    @staticmethod
    def addr(a):
        if a is None:
            return None
        if _is_torch(a):
            return a.data_ptr()
        return a.ctypes.data


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data


class PeriodEngine:
    """Owns one ``ph_ctx`` (one GPU, one stream)."""

    def __init__(self, device: int | None = None):
        self._lib = _ffi.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        n = C.c_int(0)
        _ffi.check(self._lib.ph_device_count(C.byref(n)))
        if n.value < 1:
            raise _ffi.PeriodHipError("no HIP device visible; pyperiod_amd has no CPU fallback")
        self.device = device % n.value
        ctx = C.c_void_p()
        _ffi.check(self._lib.ph_create(self.device, C.byref(ctx)))
        self._ctx = ctx
        self._bound_stream = None
        self._lock = threading.Lock()
        cu, lds = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.ph_device_info(self._ctx, C.byref(cu), C.byref(lds)))
        self.num_cu, self.lds_bytes = cu.value, lds.value

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.ph_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _prep(self, x):
        """-> (array, dtype code, W, N, flags, out-factory); binds torch's stream if needed."""
        if _is_torch(x):
            import torch

            if not x.is_cuda:
                raise ValueError("torch input must live on the GPU (or pass a numpy array)")
            if x.device.index != self.device:
                raise ValueError(f"tensor on cuda:{x.device.index}, engine on cuda:{self.device}")
            if x.dim() != 2:
                raise ValueError("expected a (W, N) batch of windows")
            x = x.contiguous()
            code = {torch.float64: _ffi.PH_F64, torch.float32: _ffi.PH_F32}.get(x.dtype)
            if code is None:
                raise TypeError(f"unsupported dtype {x.dtype}")
            mk = _Out(x)
            # torch's default stream has the handle 0 (== NULL, "own stream" in the C ABI)
            mk.stream = torch.cuda.current_stream(x.device).cuda_stream or _ffi.PH_STREAM_DEFAULT
            return x, code, x.shape[0], x.shape[1], _ffi.PH_FLAG_DEVICE, mk
        x = np.asarray(x)
        if x.ndim != 2:
            raise ValueError("expected a (W, N) batch of windows")
        if x.dtype not in _NP_DTYPES:
            x = x.astype(np.float64)
        x = np.ascontiguousarray(x)
        return x, _NP_DTYPES[x.dtype], x.shape[0], x.shape[1], 0, _Out(x)

    def _call(self, mk, W, fn, *args, check=True):
        """One library call under the engine lock: the context is bound to the caller's stream
        (torch's current stream for device tensors, the context's own stream for numpy) and the
        entry point is invoked while the lock is held, so threads sharing an engine cannot launch
        on each other's stream.  An empty batch (W == 0, e.g. a trailing rank of a sharded run)
        makes no call: the outputs are already empty."""
        if W == 0:
            return _ffi.PH_OK
        with self._lock:
            want = mk.stream
            if want != self._bound_stream:
                _ffi.check(self._lib.ph_set_stream(self._ctx, C.c_void_p(want) if want else None))
                self._bound_stream = want
            rc = fn(self._ctx, *args)
        if check:
            _ffi.check(rc)
        return rc

    @staticmethod
    def _np_dtype(code):
        return np.float64 if code == _ffi.PH_F64 else np.float32

    @staticmethod
    def _flags(trunc, orth):
        return (_ffi.PH_FLAG_TRUNC if trunc else 0) | (_ffi.PH_FLAG_ORTH if orth else 0)

    @staticmethod
    def _orth(orth, max_p):
        if not orth:
            return None, None, None, None, 0
        off, q = _factors.orth_tables(int(max_p))
        return off, q, off.ctypes.data, q.ctypes.data, int(max_p)

    def sync(self):
        _ffi.check(self._lib.ph_sync(self._ctx))

    def timer_begin(self):
        _ffi.check(self._lib.ph_timer_begin(self._ctx))

    def timer_end(self) -> float:
        ms = C.c_float(0)
        _ffi.check(self._lib.ph_timer_end(self._ctx, C.byref(ms)))
        return float(ms.value)

    def profile(self, on: bool = True):
        """Start (or stop) bracketing every kernel launch with HIP events on the stream."""
        _ffi.check(self._lib.ph_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self):
        """-> [(kernel name, milliseconds)] for every launch since profile(True)."""
        n = C.c_int(0)
        buf = (C.c_float * 256)()
        _ffi.check(self._lib.ph_profile_read(self._ctx, buf, 256, C.byref(n)))
        return [(self._lib.ph_profile_name(self._ctx, i).decode(), float(buf[i])) for i in range(min(n.value, 256))]

    def max_window(self, dtype=np.float64, trunc=False, orth=False) -> int:
        n = C.c_int(0)
        code = _NP_DTYPES[np.dtype(dtype)]
        _ffi.check(self._lib.ph_max_window(self._ctx, code, self._flags(trunc, orth), C.byref(n)))
        return n.value

    # ------------------------------------------------------------------ kernels
    def periodic_norm(self, x, p=None):
        x, code, W, N, fl, mk = self._prep(x)
        out = mk.empty((W,), np.float64)
        self._call(mk, W, self._lib.ph_periodic_norm, mk.addr(x), code, W, N, int(p) if p else 0, fl, mk.addr(out))
        return out

    def project_batch(self, x, p_list, trunc=False, orth=False, single=False):
        """out[w, k, :] = project(x[w], p_list[k]); Periods.py:142-219."""
        x, code, W, N, fl, mk = self._prep(x)
        pl, pl_addr = _i32(np.atleast_1d(p_list))
        if pl.size < 1 or pl.min() < 1:
            raise ValueError("periods must be >= 1")
        keep = self._orth(orth, pl.max())
        out = mk.empty((W, pl.size, N), self._np_dtype(code))
        flags = fl | self._flags(trunc, orth) | (_ffi.PH_FLAG_SINGLE if single else 0)
        self._call(mk, W, self._lib.ph_project_batch, mk.addr(x), code, W, N, pl_addr, pl.size, keep[2], keep[3], keep[4],
                   flags, mk.addr(out))
        return out

    def sweep(self, x, p_lo, p_hi, mode=_ffi.PH_SWEEP_NORM, trunc=False, orth=False):
        """(W, p_hi-p_lo+1) float64 sweep values; Periods.py:501-510 / :324-331."""
        x, code, W, N, fl, mk = self._prep(x)
        keep = self._orth(orth and mode != _ffi.PH_SWEEP_MAXABS, p_hi)
        out = mk.empty((W, int(p_hi) - int(p_lo) + 1), np.float64)
        self._call(mk, W, self._lib.ph_sweep, mk.addr(x), code, W, N, int(p_lo), int(p_hi), int(mode), keep[2], keep[3],
                   keep[4], fl | self._flags(trunc, orth), mk.addr(out))
        return out

    def sweep_plan_info(self, p_lo, p_hi):
        """Pass plan of the norm sweeps over [p_lo, p_hi]: (passes, periods) -- every pass reads the
        LDS-resident window once and yields one to three candidate periods."""
        n_pass, n_per = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.ph_sweep_plan_info(self._ctx, int(p_lo), int(p_hi), C.byref(n_pass), C.byref(n_per)))
        return n_pass.value, n_per.value

    def m_best_info(self, n, num=5, max_length=None, min_length=2, dtype=np.float64, trunc=False, orth=False):
        """(windows per workgroup, LDS bytes per sample and workgroup) of the step-1 kernel m_best would run:
        (2, 8) for the window-pair float screen, (1, itemsize) for the one-window fold."""
        wpw, bps = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.ph_m_best_info(self._ctx, _NP_DTYPES[np.dtype(dtype)], int(n), int(num), int(min_length),
                                            int(n // 3 if max_length is None else max_length), self._flags(trunc, orth),
                                            C.byref(wpw), C.byref(bps)))
        return wpw.value, bps.value

    def m_best_plan_info(self, n, num=5, max_length=None, min_length=2, dtype=np.float64, trunc=False, orth=False):
        """(passes, periods) of one sweep of the step-1 kernel m_best would run (the window-pair kernel takes the
        periods up to 64 in chains, so it needs fewer passes than sweep_plan_info reports for the fp64 sweeps)."""
        n_pass, n_per = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.ph_m_best_plan_info(self._ctx, _NP_DTYPES[np.dtype(dtype)], int(n), int(num), int(min_length),
                                                 int(n // 3 if max_length is None else max_length),
                                                 self._flags(trunc, orth), C.byref(n_pass), C.byref(n_per)))
        return n_pass.value, n_per.value

    def m_best(self, x, num=5, max_length=None, min_length=2, gamma=False, trunc=False, orth=False, want_sweeps=False):
        """-> periods (W,num) uint32, powers (W,num) f64, bases (W,num,N), status (W) int32
        [, n_sweeps (W) int32 when want_sweeps]."""
        x, code, W, N, fl, mk = self._prep(x)
        if max_length is None:
            max_length = N // 3
        max_length, min_length, num = int(max_length), int(min_length), int(num)
        keep = self._orth(orth, max_length)
        foff, fq = _factors.factor_tables(max(max_length, 1))
        periods = mk.empty((W, num), np.uint32)
        powers = mk.empty((W, num), np.float64)
        bases = mk.empty((W, num, N), self._np_dtype(code))
        status = mk.empty((W,), np.int32)
        sweeps = mk.empty((W,), np.int32) if want_sweeps else None
        self._call(mk, W, self._lib.ph_m_best, mk.addr(x), code, W, N, num, min_length, max_length, 1 if gamma else 0,
                   keep[2], keep[3], foff.ctypes.data, fq.ctypes.data, max(max_length, 1),
                   fl | self._flags(trunc, orth), mk.addr(periods), mk.addr(powers), mk.addr(bases), mk.addr(status),
                   mk.addr(sweeps))
        if want_sweeps:
            return periods, powers, bases, status, sweeps
        return periods, powers, bases, status

    def small_to_large(self, x, thresh=0.1, n_periods=None, trunc=False, orth=False, cap=16, want_bases=True,
                       nosync=False):
        """-> counts (W), periods (W,cap) int32, powers (W,cap), bases (W,cap,N)|None, status (W).
        A window that accepts more than `cap` periods makes the library return PH_E_CAP -- for
        numpy and for device tensors alike (the library reads one device word back) -- and the
        call is repeated with the capacity the batch needs.  nosync=True (device tensors only)
        skips that read-back: the call stays asynchronous and the caller must look at `status`
        (PH_ST_CAP) / `counts` itself."""
        x, code, W, N, fl, mk = self._prep(x)
        if n_periods is None:
            n_periods = N // 2
        n_periods = int(n_periods)
        keep = self._orth(orth, max(n_periods, 1))
        fl |= _ffi.PH_FLAG_NOSYNC if (nosync and mk.torch) else 0
        while True:
            counts = mk.empty((W,), np.int32)
            periods = mk.empty((W, cap), np.int32)
            powers = mk.empty((W, cap), np.float64)
            bases = mk.empty((W, cap, N), self._np_dtype(code)) if want_bases else None
            status = mk.empty((W,), np.int32)
            rc = self._call(mk, W, self._lib.ph_small_to_large, mk.addr(x), code, W, N, float(thresh), n_periods,
                            keep[2], keep[3], keep[4], fl | self._flags(trunc, orth), int(cap), mk.addr(counts),
                            mk.addr(periods), mk.addr(powers), mk.addr(bases), mk.addr(status), check=False)
            if rc == _ffi.PH_E_CAP:
                cap = int(counts.max())
                continue
            _ffi.check(rc)
            return counts, periods, powers, bases, status

    def best_correlation(self, x, num=5, max_length=None, ratio=0.01, trunc=False, orth=False):
        x, code, W, N, fl, mk = self._prep(x)
        if max_length is None:
            max_length = N // 3
        max_length, num = int(max_length), int(num)
        keep = self._orth(orth, max(max_length, 1))
        periods = mk.empty((W, num), np.uint32)
        norms = mk.empty((W, num), np.float64)
        bases = mk.empty((W, num, N), self._np_dtype(code))
        status = mk.empty((W,), np.int32)
        self._call(mk, W, self._lib.ph_best_correlation, mk.addr(x), code, W, N, num, max_length, float(ratio), keep[2],
                   keep[3], keep[4], fl | self._flags(trunc, orth), mk.addr(periods), mk.addr(norms), mk.addr(bases),
                   mk.addr(status))
        return periods, norms, bases, status

    def best_frequency(self, x, win_size=None, num=5, trunc=False, orth=False):
        """Periods.best_frequency over a batch (Periods.py:351-398): periods (W, num) uint32, powers
        (W, num), bases (W, num, N), status (W) -- PH_ST_NO_PERIOD where the reference raises."""
        x, code, W, N, fl, mk = self._prep(x)
        win_size = N if win_size is None else int(win_size)
        num = int(num)
        keep = self._orth(orth, 2 * max(win_size, 1))
        periods = mk.empty((W, num), np.uint32)
        powers = mk.empty((W, num), np.float64)
        bases = mk.empty((W, num, N), self._np_dtype(code))
        status = mk.empty((W,), np.int32)
        for w0 in range(0, W, 65535):  # the spectrum kernel's grid.y holds the window index
            w1 = min(W, w0 + 65535)
            self._call(mk, w1 - w0, self._lib.ph_best_frequency, mk.addr(x[w0:w1]), code, w1 - w0, N, win_size, num,
                       keep[2], keep[3], keep[4], fl | self._flags(trunc, orth), mk.addr(periods[w0:w1]),
                       mk.addr(powers[w0:w1]), mk.addr(bases[w0:w1]), mk.addr(status[w0:w1]))
        return periods, powers, bases, status

    def ramanujan_norms(self, x, q_lo=2, q_hi=None):
        """(W, q_hi+1) float64; RamanujanPeriods.py:67-86."""
        x, code, W, N, fl, mk = self._prep(x)
        if not q_hi:
            q_hi = N // 3
        out = mk.empty((W, int(q_hi) + 1), np.float64)
        self._call(mk, W, self._lib.ph_ramanujan_norms, mk.addr(x), code, W, N, int(q_lo), int(q_hi), fl, mk.addr(out))
        return out

    def dict_project(self, x, basis):
        """RamanujanPeriods.project(x, basis) for an arbitrary dictionary (numpy only)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        basis = np.ascontiguousarray(basis, dtype=np.float64)
        if x.ndim != 1 or basis.ndim != 2 or basis.shape[1] != x.size:
            raise ValueError("x must be (N,) and basis (rows, N)")
        out = np.empty(basis.shape, dtype=np.float32)
        self._call(_Out(x), basis.shape[0], self._lib.ph_dict_project, x.ctypes.data, basis.ctypes.data, basis.shape[0],
                   x.size, 0, out.ctypes.data)
        return out

    def qo_find_periods(self, x, num, thresh, min_length=2, max_length=None, kcap=512):
        """QOPeriods.find_periods (plain projection, update_weights=True, default test function)
        for a batch.  -> periods (W,num) u32, norms (W,num), keeps (W,num) i32, counts (W,2) i32,
        weights (W,kcap) f64, residual (W,N), status (W)."""
        x, code, W, N, fl, mk = self._prep(x)
        if max_length is None:
            max_length = N // 3
        num = int(num)
        periods = mk.empty((W, num), np.uint32)
        norms = mk.empty((W, num), np.float64)
        keeps = mk.empty((W, num), np.int32)
        counts = mk.empty((W, 2), np.int32)
        weights = mk.empty((W, int(kcap)), np.float64)
        resid = mk.empty((W, N), self._np_dtype(code))
        status = mk.empty((W,), np.int32)
        self._call(mk, W, self._lib.ph_qo_find_periods, mk.addr(x), code, W, N, num, float(thresh), int(min_length),
                   int(max_length), int(kcap), fl, mk.addr(periods), mk.addr(norms), mk.addr(keeps), mk.addr(counts),
                   mk.addr(weights), mk.addr(resid), mk.addr(status))
        return periods, norms, keeps, counts, weights, resid, status

    def qo_feasible(self, n, dtype=np.float64, kcap=512, max_length=None) -> bool:
        """Whether ph_qo_find_periods can run a window of n samples with `kcap` dictionary rows
        (bookkeeping + the work vectors of the conjugate-gradient solve must fit the workgroup's LDS)."""
        ok = C.c_int(0)
        code = _NP_DTYPES[np.dtype(dtype)]
        _ffi.check(self._lib.ph_qo_feasible(self._ctx, code, int(n), int(max_length if max_length is not None else n // 3),
                                            int(kcap), C.byref(ok)))
        return bool(ok.value)

    def orth_powers(self, x, max_p=None, normalize=False, want_autocorr=False, want_eq3=False):
        """Orthogonal period powers (QOPeriods.get_best_period_orthogonal(return_powers=True)).
        -> pows (W, max_p) [, autocorr (W, N)] [, eq3 (W, max_p)]."""
        x, code, W, N, fl, mk = self._prep(x)
        if max_p is None:
            max_p = N // 2
        max_p = int(max_p)
        pows = mk.empty((W, max_p), np.float64)
        ac = mk.empty((W, N), np.float64) if want_autocorr else None
        e3 = mk.empty((W, max_p), np.float64) if want_eq3 else None
        self._call(mk, W, self._lib.ph_orth_powers, mk.addr(x), code, W, N, max_p, 1 if normalize else 0, fl,
                   mk.addr(ac), mk.addr(e3), mk.addr(pows))
        out = (pows,)
        if want_autocorr:
            out += (ac,)
        if want_eq3:
            out += (e3,)
        return out if len(out) > 1 else pows

    def fold_sums(self, x, p_list, keep):
        """W = A x for natural-basis rows (QOPeriods.py:782): (W, sum(keep)) float64."""
        x, code, W, N, fl, mk = self._prep(x)
        pl, pl_addr = _i32(np.atleast_1d(p_list))
        kp, kp_addr = _i32(np.atleast_1d(keep))
        out = mk.empty((W, int(kp.sum())), np.float64)
        self._call(mk, W, self._lib.ph_fold_sums, mk.addr(x), code, W, N, pl_addr, kp_addr, pl.size, fl, mk.addr(out))
        return out

    def tile_sum(self, wts, n, p_list, keep, dtype=np.float64):
        """Reconstruction A^T w (QOPeriods.py:795): (W, n)."""
        wts2, code, W, S, fl, mk = self._prep(wts)
        if code != _ffi.PH_F64:
            raise TypeError("weights must be float64")
        pl, pl_addr = _i32(np.atleast_1d(p_list))
        kp, kp_addr = _i32(np.atleast_1d(keep))
        if int(kp.sum()) != S:
            raise ValueError("weights row length must equal sum(keep)")
        ocode = _NP_DTYPES[np.dtype(dtype)]
        out = mk.empty((W, int(n)), self._np_dtype(ocode))
        self._call(mk, W, self._lib.ph_tile_sum, mk.addr(wts2), W, int(n), pl_addr, kp_addr, pl.size, ocode, fl, mk.addr(out))
        return out


_default = None
_default_lock = threading.Lock()


def default_engine() -> PeriodEngine:
    """Process-wide engine on cuda:LOCAL_RANK (one process per GPU)."""
    global _default
    with _default_lock:
        if _default is None:
            _default = PeriodEngine()
        return _default
