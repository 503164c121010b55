"""Drop-in ``QOPeriods`` (quadratic-optimisation period finder) on the MI355X engine.

Mirrors the reference surface (pyPeriod/QOPeriods.py:148-1310).  The v1 reference class cannot
even be constructed (QOPeriods.py:190) and only its non-orthogonal ``find_periods`` branch runs
(SURVEY.md section 0); that branch is what is implemented here:

  * plain projection, default test function: the whole greedy loop (gamma sweep, phi-mass row
    bookkeeping, right-hand side by folds, matrix-free conjugate-gradient solve, reconstruction,
    residual) runs in ONE kernel launch per window batch -> ph_qo_find_periods
  * other settings (custom test_function, update_weights=False, trunc, window, Ramanujan basis):
    the loop is driven from the host with the heavy pieces on the GPU -- the sweep (ph_sweep,
    QOPeriods.py:470-478), W = A x and A A^T as folds (ph_fold_sums, :781-782), A^T w
    (ph_tile_sum, :795) -- and the small dense solve on host LAPACK like the reference (:794).
"""

from __future__ import annotations

import warnings

import numpy as np

from . import _ffi
from ._factors import PRIMES, get_factors, get_primes, phi  # noqa: F401
from .engine import default_engine
from .Periods import Periods, _as_window, rms


def flatten(t: list) -> list:
    return [item for sublist in t for item in sublist]


def normalize(x, level: int = 1):
    """Scale so that max |x| == level (QOPeriods.py:119-145)."""
    x = np.asarray(x)
    return (x / np.max(np.abs(x))) * level


def ramanujan_sum(q: int) -> np.ndarray:
    """c_q(n) for n < q as exact integers: c_q(n) = sum_{d | gcd(n,q)} mu(q/d) d.  The reference
    evaluates the same numbers through complex exponentials (QOPeriods.py:1035-1044) and carries
    ~1e-12 of rounding noise on top of these integers."""
    q = int(q)
    mu = np.ones(q + 1, dtype=np.int64)
    is_comp = np.zeros(q + 1, dtype=bool)
    for i in range(2, q + 1):
        if not is_comp[i]:
            is_comp[2 * i :: i] = True
            mu[i::i] *= -1
            mu[i * i :: i * i] = 0
    n = np.arange(q)
    g = np.gcd(n, q)
    out = np.zeros(q, dtype=np.int64)
    for d in range(1, q + 1):
        if q % d == 0 and mu[q // d] != 0:
            out[g % d == 0] += mu[q // d] * d
    return out


class QOPeriods(Periods):
    PRIMES = PRIMES  # QOPeriods.py:149-151

    def __init__(self, basis_type="natural", trunc_to_integer_multiple=False, orthogonalize=False):
        super().__init__(trunc_to_integer_multiple, orthogonalize)
        # attribute list of QOPeriods.py:191-199
        self._output = None
        self._basis_type = basis_type
        self._verbose = False
        self._k = 0
        self._window = False
        self._output_bases = None
        self._container = []

    # ------------------------------------------------------------------ detection
    def find_periods(self, data, num=None, thresh=None, min_length=2, max_length=None, update_weights=True, **kwargs):
        """Greedy period selection with re-solved weights (QOPeriods.py:313-596).
        Returns ``(dict(periods, norms, subspaces, weights, basis_dictionary), residual)``.

        ``orthogonalize=True``: the v1 reference dies on this branch (``best_base`` is never
        assigned, QOPeriods.py:427-448).  Offered here as its commented-out lines intend: the
        period is chosen by the orthogonal (Muresan-Parks) powers -- ``get_best_period_orthogonal``
        on the device -- and its norm is that of the orthogonalised projection of the residual
        (QOPeriods.py:443-448); the solve is the same as in the plain branch."""
        data = _as_window(data)
        N = len(data)
        if max_length is None:
            max_length = int(np.floor(N / 3))
        if num is None:
            num = N
        if np.sum(np.abs(data)) <= 1e-16:  # QOPeriods.py:394-406
            self._output = {
                "periods": np.array([1]),
                "norms": np.array([0]),
                "subspaces": np.ones((1, N)),
                "weights": np.array([0]),
                "basis_dictionary": {"1": N},
            }
            return (self._output, np.zeros(N))
        custom_test = kwargs.get("test_function")
        windowed = not (self.window is None or self.window is False)
        on_device = (
            update_weights and custom_test is None and thresh is not None and not self._orthogonalize
            and not self._trunc_to_integer_multiple and self._basis_type == "natural" and not windowed
        )
        if on_device:
            done = self._find_periods_device(default_engine(), data, N, num, thresh, min_length, max_length)
            if done is not None:
                return done
        return self._find_periods_host(data, N, num, thresh, min_length, max_length, update_weights, custom_test)

    def _strongest_period(self, eng, res, found, min_length, max_length, update_weights):
        """(period, gamma norm) of the residual `res`; period 0 = stop (QOPeriods.py:425-478)."""
        if not self._orthogonalize:  # plain gamma sweep, first maximum == strict '>' scan (:470-478)
            vals = eng.sweep(res[None, :], min_length, max_length, _ffi.PH_SWEEP_NORM_GAMMA, self._trunc_to_integer_multiple, False)[0]
            order = np.where(np.isnan(vals), -np.inf, vals)
            k = int(np.argmax(order))
            return (min_length + k, vals[k]) if order[k] > 0 else (0, 0)
        pows = eng.orth_powers(res[None, :], int(max_length), True)[0]
        if update_weights:  # :435-448
            best = int(np.argmax(pows))
            best = best if best > 0 else 1
        else:  # strongest power not found yet (:450-460)
            best = next((int(q) for q in np.argsort(-pows, kind="stable") if q not in found), 0)
        if best < 1:
            return 0, 0
        base = eng.project_batch(res[None, :], [best], self._trunc_to_integer_multiple, True)[0, 0]
        return best, eng.periodic_norm(base[None, :], best)[0]

    def _find_periods_host(self, data, N, num, thresh, min_length, max_length, update_weights, custom_test):
        """Host-driven greedy loop for the variants the single-launch kernel does not cover (custom
        test function, update_weights=False, trunc, window, Ramanujan basis, orthogonal selection).
        Heavy pieces stay on the GPU: the sweep (ph_sweep) or orthogonal powers (ph_orth_powers),
        W = A x and A A^T as folds (ph_fold_sums), A^T w (ph_tile_sum); the small dense solve uses
        host LAPACK like the reference (QOPeriods.py:794)."""
        eng = default_engine()
        keep_going = custom_test if custom_test is not None else (lambda _self, x, y: rms(y) > rms(data) * thresh)
        found = np.zeros(num, dtype=np.uint32)
        gnorm = np.zeros(num)
        state = {"A": np.empty((0, N)), "dims": {}, "w": np.array([]), "recon": None}
        res = data.copy()
        result = {"periods": [], "norms": [], "subspaces": [], "weights": [], "basis_dictionary": {}}

        def resolve(active):
            # update_weights: dictionary of all periods so far, weights re-solved against the data
            # (:598-643); otherwise only the newest period's rows are fitted to the residual (:645-714)
            if update_weights:
                state["A"], state["dims"], state["w"], state["recon"] = self._update_weights(data, N, active)
            else:
                state["A"], state["dims"], state["w"], state["recon"] = self._dont_update_weights(
                    res, N, active, state["w"], state["A"], state["dims"]
                )

        def report(active, count):
            return {
                "periods": active[:count],
                "norms": gnorm[:count],
                "subspaces": state["A"],
                "weights": state["w"],
                "basis_dictionary": state["dims"],
            }

        active = found[:0]
        for i in range(num):
            if i > 0 and not keep_going(self, data, state["recon"]):
                # the period added last broke the test: weights / dictionary of everything found are kept,
                # the period list drops its last entry (:560-594)
                resolve(active)
                result = report(active, len(active) - 1)
                self._output_bases = result
                break
            p, g = self._strongest_period(eng, res, found, min_length, max_length, update_weights)
            if self._orthogonalize and p < 1:
                break  # :441-442
            found[i], gnorm[i] = p, g
            if self._verbose:
                print(f"New period: {p}")
            active = found[found > 0]
            try:
                resolve(active)
            except np.linalg.LinAlgError:  # singular dictionary: keep the previous result (:552-559)
                break
            res = (data - state["recon"]) if update_weights else (res - state["recon"])
            result = report(active, len(active))
            self._output_bases = result
        return (result, res)

    def _find_periods_device(self, eng, data, N, num, thresh, min_length, max_length):
        """The whole greedy loop in one kernel launch (ph_qo_find_periods); None when the window /
        dictionary does not fit the kernel's LDS layout or workspace -- the host-driven loop then runs."""
        # a block adds at most max_length rows: start with room for all of them when that fits
        bound = int(num) * int(max_length if max_length is not None else N // 3)
        kcap = min(2048, max(64, -(-bound // 64) * 64)) if bound <= 2048 else 512
        while kcap > 64 and not eng.qo_feasible(N, np.float64, kcap, max_length):
            kcap //= 2
        if not eng.qo_feasible(N, np.float64, kcap, max_length):
            return None
        while True:
            per, nrm, keeps, counts, wts, resid, st = eng.qo_find_periods(data[None, :], num, thresh, min_length, max_length, kcap)
            if st[0] == _ffi.PH_ST_CAP and kcap < 2048 and eng.qo_feasible(N, np.float64, 2 * kcap, max_length):
                kcap *= 2
                continue
            break
        if st[0] != _ffi.PH_ST_OK:
            return None  # dictionary larger than the device workspace: host-driven loop
        n_report, n_blocks = int(counts[0, 0]), int(counts[0, 1])
        if n_blocks == 0:
            return None
        blocks = [(int(per[0, b]), int(keeps[0, b])) for b in range(n_blocks)]
        rows = np.vstack([self.Pp(q, N, k, self._basis_type) for q, k in blocks])
        result = {
            "periods": per[0, :n_report].copy(),
            "norms": nrm[0, :n_report].copy(),
            "subspaces": rows,
            "weights": wts[0, : rows.shape[0]].copy(),
            "basis_dictionary": {str(q): k for q, k in blocks},
        }
        self._output_bases = result
        return (result, resid[0].copy())

    def _solve_structured(self, x, basis_matrix, dictionary):
        """solve_quadratic for a natural-basis dictionary without touching the dense matrix
        on the compute side: W = A x and A A^T are folds, A^T w is a tile-sum."""
        if self._basis_type != "natural":
            return self.solve_quadratic(x, basis_matrix, window=self.window)
        eng = default_engine()
        p_list = [int(q) for q in dictionary.keys()]
        keep = [int(v) for v in dictionary.values()]
        win = self.window
        if win is None or win is False:
            rhs = eng.fold_sums(x[None, :], p_list, keep)[0]
            gram = eng.fold_sums(basis_matrix, p_list, keep)  # row (q,j) folded by p == (A A^T)^T
        else:
            rhs = eng.fold_sums((x * win)[None, :], p_list, keep)[0]
            gram = eng.fold_sums(basis_matrix * win, p_list, keep)
        weights = np.linalg.solve(gram.T, rhs)  # QOPeriods.py:794 (raises LinAlgError when singular)
        recon = eng.tile_sum(weights[None, :], x.size, p_list, keep)[0]
        return weights, recon

    def _update_weights(self, data, N, nonzero_periods):
        """QOPeriods.py:598-643."""
        basis_matricies, basis_dictionary = self.get_subspaces(nonzero_periods, N)
        output_weights, reconstruction = self._solve_structured(data, basis_matricies, basis_dictionary)
        return (basis_matricies, basis_dictionary, output_weights, reconstruction)

    def _dont_update_weights(self, data, N, nonzero_periods, output_weights, basis_matricies, basis_dictionary):
        """QOPeriods.py:645-714 (the v1 reference overflows here under numpy 2: uint32 periods
        reach Pp_column; periods are converted to int first)."""
        last = int(nonzero_periods[-1])
        keep = last
        factors = get_factors(last)
        existing = set()
        for p in nonzero_periods[:-1]:
            existing = existing.union(get_factors(int(p)))
        for f in sorted(existing.intersection(factors)):
            keep -= phi(f)
        basis_matrix = self.Pp(last, N, keep=keep, type=self._basis_type)
        weights, reconstruction = self._solve_structured(data, basis_matrix, {str(last): basis_matrix.shape[0]})
        basis_dictionary.update({str(last): keep})
        basis_matricies = np.vstack((basis_matricies, basis_matrix))
        output_weights = np.concatenate((output_weights, weights))
        return (basis_matricies, basis_dictionary, output_weights, reconstruction)

    # ------------------------------------------------------------------ linear algebra
    @staticmethod
    def solve_quadratic(x, A, type: str = "solve", window=None, k: int = 0):
        """Generic dense form (QOPeriods.py:743-805): A' = A A^T, W = A x, solve, A^T w.
        The two dense products are plain library GEMMs and run through rocBLAS
        (torch.matmul on the GPU); the small solve uses host LAPACK like the reference."""
        import torch

        eng = default_engine()
        dev = torch.device("cuda", eng.device)
        At = torch.as_tensor(np.ascontiguousarray(A, dtype=np.float64), device=dev)
        xt = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)
        Aw = At if (window is None or window is False) else At * torch.as_tensor(np.asarray(window, dtype=np.float64), device=dev)
        A_prime = (Aw @ At.T).cpu().numpy()
        W = (Aw @ xt).cpu().numpy()
        if type == "solve":
            output = np.linalg.solve(A_prime, W)
        else:
            if type != "lstsq":
                warnings.warn("type ({}) unrecognized. Defaulting to lstsq.".format(type))
            output = np.linalg.lstsq(A_prime, W, rcond=None)[0]
        reconstructed = (At.T @ torch.as_tensor(output, device=dev)).cpu().numpy()
        return (output, reconstructed)

    def get_subspaces(self, Q, N: int):
        """Stacked natural-basis rows and the {period: rows kept} dictionary
        (QOPeriods.py:807-852)."""
        old_dimensionality = 0
        d = {}
        R = set()
        for q in Q:
            R = R.union(get_factors(int(q)))
            s = int(np.sum([phi(r) for r in R]))
            d[str(q)] = s - old_dimensionality
            old_dimensionality = s
        blocks = [self.Pp(int(q), N, keep, self._basis_type) for q, keep in d.items()]
        A = np.vstack(blocks) if blocks else np.array([]).reshape((0, N))
        return (A, d)

    def compute_reconstruction(self, x, periods, type: str = "lstsq", window=None):
        """QOPeriods.py:1054-1116."""
        basis_matricies, basis_dictionary = self.get_subspaces(periods, len(x))
        try:
            output_weights, reconstruction = self.solve_quadratic(x, basis_matricies, window=window, type=type)
        except np.linalg.LinAlgError:
            return None
        output_bases = {
            "periods": periods,
            "subspaces": basis_matricies,
            "weights": output_weights,
            "basis_dictionary": basis_dictionary,
        }
        return (reconstruction, output_bases)

    def get_periods(self, weights, dictionary, decomp_type="row reduction"):
        """QOPeriods.py:719-741 -- raises TypeError in the v1 reference (positional `_k` lands
        in `type`); out of the hot-path scope."""
        raise NotImplementedError("QOPeriods.get_periods is outside the accelerated path (broken in reference v1)")

    @staticmethod
    def concatenate_periods(weights, dictionary):
        """QOPeriods.py:854-887."""
        read_idx = 0
        output = []
        for q, r in dictionary.items():
            v = np.zeros(int(q))
            v[0:r] = weights[read_idx : read_idx + r]
            read_idx += r
            output.append(v)
        return np.array(flatten(output))

    # ------------------------------------------------------------------ dictionaries (host tables)
    @staticmethod
    def Pp(p: int, N: int = 1, keep: int = None, type: str = "natural") -> np.ndarray:
        """Natural (indicator) or Ramanujan basis rows (QOPeriods.py:940-974)."""
        p, N = int(p), int(N)
        if type == "natural":
            matrix = (np.arange(N)[None, :] % p == np.arange(p)[:, None]).astype(np.float64)
        elif type == "ramanujan":
            reps = int(np.ceil(N / p))
            matrix = np.stack([QOPeriods.Cq(p, i, reps, "real")[:N] for i in range(p)])
        else:
            matrix = np.zeros((p, N))
        return matrix[:keep] if keep else matrix

    @staticmethod
    def Pp_column(p: int, s: int, repetitions: int = 1):
        """[.., 1 at (index - s) % p == 0, ..] tiled (QOPeriods.py:976-1003)."""
        p, s = int(p), int(s)
        vec = ((np.arange(p) - s) % p == 0).astype(np.float64)
        return np.tile(vec, repetitions)

    @staticmethod
    def Cq(q: int, s: int = 0, repetitions: int = 1, type: str = "real"):
        """Ramanujan sum c_q rolled by s and tiled (QOPeriods.py:1005-1052)."""
        vec = np.tile(np.roll(ramanujan_sum(q).astype(np.float64), s), repetitions)
        return vec.astype(complex) if type == "complex" else vec

    # ------------------------------------------------------------------ orthogonal period powers
    def auto_corr(self, x, k):
        """sum_n x[n] x[n+k] (QOPeriods.py:1151-1173)."""
        x = _as_window(x)
        return np.float64(default_engine().orth_powers(x[None, :], 2, want_autocorr=True)[1][0, int(k)])

    def eq_3(self, x, P):
        """Equation 3 of Muresan & Parks (QOPeriods.py:1122-1149)."""
        x = _as_window(x)
        P = int(P)
        return np.float64(default_engine().orth_powers(x[None, :], max(P + 1, 2), want_eq3=True)[1][0, P])

    def get_best_period_orthogonal(self, x, max_p=None, normalize=False, return_powers=False):
        """Strongest period by orthogonal (factor-subtracted) powers (QOPeriods.py:1175-1232)."""
        x = _as_window(x)
        if max_p is None:
            max_p = len(x) // 2
        pows = default_engine().orth_powers(x[None, :], int(max_p), normalize)[0]
        if return_powers:
            return pows
        best = int(np.argmax(pows))
        return best if best > 0 else 1  # Q[argmax] - 1 == argmax (QOPeriods.py:1227-1232)

    # ------------------------------------------------------------------ properties (QOPeriods.py:1237-1310)
    @property
    def basis_type(self):
        return self._basis_type

    @basis_type.setter
    def basis_type(self, value):
        self._basis_type = value

    @property
    def verbose(self):
        return self._verbose

    @verbose.setter
    def verbose(self, value):
        self._verbose = value

    @property
    def k(self):
        return self._k

    @k.setter
    def k(self, value):
        self._k = value

    @property
    def window(self):
        return self._window

    @window.setter
    def window(self, value):
        self._window = value

    @property
    def output_bases(self):
        return self._output_bases
