"""
Many independent scipy L-BFGS-B minimisations advanced side by side in ONE Python thread.

The reference aligns every projection with `scipy.optimize.minimize(cost_xzab, ..., method='L-BFGS-B', jac=gradient_xzab, bounds=...)`
(examples/align_rigid.py:44-49).  scipy's driver (`scipy/optimize/_lbfgsb_py.py::_minimize_lbfgsb`) is a reverse-communication loop
around the compiled core `_lbfgsb.setulb`: the core returns with task FG whenever it wants f and g at a point.  `Lbfgsb` below is that
loop turned inside out -- `advance()` runs the core until it asks for an evaluation (or stops), `feed(f, g)` hands the values in -- so a
caller can hold hundreds of optimisers, collect their pending points, evaluate them in one GPU launch and feed the results back,
without a thread per optimiser (round 3 ran 256 Python threads in lock step: a third of the wall time was GIL hand-overs).
Same core, same arguments, same bookkeeping (nfev, maxiter / maxfun tests, bound clipping of x0): the iterates are scipy's own.

`AVAILABLE` is False when this scipy's private core does not have the layout this module was written against (scipy 1.15: integer
`task` arrays); `self_test()` additionally runs a small bounded problem through both drivers and requires identical iterates.
Callers (alignment.py) fall back to one `optimize.minimize` per worker thread then.
"""
import numpy as np

try:
    from scipy.optimize import _lbfgsb as _core
    from scipy.optimize import _lbfgsb_py as _py
    AVAILABLE = hasattr(_core, "setulb") and getattr(_py, "status_messages", {}).get(3) == "FG" and getattr(_py, "status_messages", {}).get(1) == "NEW_X"
except Exception:      # noqa: BLE001
    _core = _py = None
    AVAILABLE = False

_T_NEW_X, _T_FG, _T_CONVERGENCE, _T_STOP = 1, 3, 4, 5


class Lbfgsb(object):
    """One L-BFGS-B minimisation in reverse communication (scipy/optimize/_lbfgsb_py.py::_minimize_lbfgsb, same defaults)."""

    def __init__(self, x0, bounds=None, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxiter=15000, maxls=20, **unknown):
        if unknown:      # scipy warns about option keys it does not know (OptimizeWarning); a misspelled `maxiter` must not pass silently here either
            raise TypeError("L-BFGS-B reverse-communication driver: unsupported option(s) %s (supported: maxcor, ftol, gtol, maxfun, maxiter, maxls)"
                            % sorted(unknown))
        x0 = np.asarray(x0, np.float64).ravel()
        n = x0.size
        self.m, self.n = int(maxcor), n
        self.factr = ftol / np.finfo(float).eps
        self.pgtol = gtol
        self.maxfun, self.maxiter, self.maxls = maxfun, maxiter, int(maxls)
        if not self.maxls > 0:
            raise ValueError("maxls must be positive.")
        self.nbd = np.zeros(n, np.int32)
        self.low = np.zeros(n, np.float64)
        self.up = np.zeros(n, np.float64)
        if bounds is not None:
            if len(bounds) != n:
                raise ValueError("length of x0 != length of bounds")
            lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], np.float64)
            hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], np.float64)
            if (lo > hi).any():
                raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
            x0 = np.clip(x0, lo, hi)
            code = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}
            for i in range(n):
                has_l, has_u = not np.isinf(lo[i]), not np.isinf(hi[i])
                if has_l:
                    self.low[i] = lo[i]
                if has_u:
                    self.up[i] = hi[i]
                self.nbd[i] = code[has_l, has_u]
        m = self.m
        self.x = np.array(x0, np.float64)
        self.f = np.array(0.0, np.int32)                  # as scipy passes it before the first evaluation
        self.g = np.zeros((n,), np.float64)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, np.int32)
        self.task = np.zeros(2, np.int32)
        self.ln_task = np.zeros(2, np.int32)
        self.lsave = np.zeros(4, np.int32)
        self.isave = np.zeros(44, np.int32)
        self.dsave = np.zeros(29, np.float64)
        self.nit = 0
        self.nfev = 0
        self.done = False

    def advance(self):
        """Run the core until it wants f, g at self.x (returns True: evaluate a COPY of self.x and feed()) or has stopped (False)."""
        while True:
            _core.setulb(self.m, self.x, self.low, self.up, self.nbd, self.f, self.g, self.factr, self.pgtol, self.wa, self.iwa, self.task,
                         self.lsave, self.isave, self.dsave, self.maxls, self.ln_task)
            t = self.task[0]
            if t == _T_FG:
                return True
            if t == _T_NEW_X:
                self.nit += 1
                if self.nit >= self.maxiter:
                    self.task[0], self.task[1] = _T_STOP, 504
                elif self.nfev > self.maxfun:
                    self.task[0], self.task[1] = _T_STOP, 502
            else:
                self.done = True
                return False

    def feed(self, f, g):
        self.f = float(f)
        self.g = np.array(g, np.float64)                  # a fresh float64 array, as `g.astype(np.float64)` gives scipy's loop
        self.nfev += 1

    @property
    def status(self):
        if self.task[0] == _T_CONVERGENCE:
            return 0
        return 1 if (self.nfev > self.maxfun or self.nit >= self.maxiter) else 2

    @property
    def message(self):
        return _py.status_messages[int(self.task[0])] + ": " + _py.task_messages[int(self.task[1])]


def minimize_many(fun_batch, x0, bounds=None, options=None, overlap=None):
    """Minimise len(x0) independent problems.  fun_batch(ids, X) -> (f[len(ids)], g[len(ids), n]) evaluates the problems `ids` at the
    rows of X.  With `overlap` (a callable taking a zero-argument function and returning an object with .result(), e.g. a one-worker
    executor's submit) the problems are split into two halves that take turns: one half is being evaluated while this thread runs
    the other half's optimiser steps; EVERY evaluation then goes through `overlap` (the tail and small problem counts too), so one
    thread owns the device for the whole pass.  Returns (x, fun, nfev, status) arrays.

    Supported scipy range: the private core `scipy.optimize._lbfgsb.setulb` with integer task arrays (scipy 1.15.x; `AVAILABLE` and
    `self_test()` decide at run time, nothing is assumed from the version string)."""
    opts = dict(options or {})
    for k in ("disp", "iprint", "eps", "callback", "finite_diff_rel_step"):      # accepted by scipy's front end, meaningless with jac=True / here
        opts.pop(k, None)
    x0 = np.asarray(x0, np.float64)
    n_prob = x0.shape[0]
    st = [Lbfgsb(x0[i], bounds=bounds, **opts) for i in range(n_prob)]

    def step(ids):
        """advance the optimisers `ids`; returns the ids that now wait for an evaluation"""
        return [i for i in ids if st[i].advance()]

    def points(ids):
        return np.array([st[i].x for i in ids], np.float64).reshape(len(ids), -1)

    def give(ids, res):
        f, g = res
        for k, i in enumerate(ids):
            st[i].feed(f[k], g[k])

    def evaluate_now(w):
        """blocking evaluation -- on the helper thread too when there is one (one thread owns the device)"""
        X = points(w)
        return fun_batch(w, X) if overlap is None else overlap(lambda: fun_batch(w, X)).result()

    ids = list(range(n_prob))
    if overlap is None or n_prob < 64:
        wait = step(ids)
        while wait:
            give(wait, evaluate_now(wait))
            wait = step(wait)
    else:
        wa, wb = step(ids[0::2]), step(ids[1::2])
        fut_a = overlap(lambda w=wa, X=points(wa): fun_batch(w, X)) if wa else None
        while wa or wb:
            # A is on the device; B's evaluation goes next, then this thread steps A's optimisers while B is being evaluated
            if len(wa) + len(wb) < 64:                 # the tail: too few left for two worthwhile launches -- one population
                if fut_a is not None:
                    give(wa, fut_a.result())
                    wa = step(wa)
                wait = wa + wb
                while wait:
                    give(wait, evaluate_now(wait))
                    wait = step(wait)
                break
            fut_b = overlap(lambda w=wb, X=points(wb): fun_batch(w, X)) if wb else None
            if fut_a is not None:
                give(wa, fut_a.result())
                wa = step(wa)                              # while B is on the device
            fut_a = overlap(lambda w=wa, X=points(wa): fun_batch(w, X)) if wa else None
            if fut_b is not None:
                give(wb, fut_b.result())
                wb = step(wb)                              # while A is on the device
    x = np.array([s.x for s in st]).reshape(n_prob, -1)
    return x, np.array([float(s.f) for s in st]), np.array([s.nfev for s in st], np.int64), np.array([s.status for s in st], np.int64)


def self_test():
    """Identical iterates to scipy.optimize.minimize on a small bounded problem (x, fun, nfev, nit bit for bit)?"""
    if not AVAILABLE:
        return False
    try:
        from scipy import optimize

        def fg(p):
            p = np.asarray(p, np.float64)
            f = 100.0 * (p[1] - p[0] ** 2) ** 2 + (1 - p[0]) ** 2 + 3.0 * (p[2] - 0.3) ** 2 + (p[3] + 0.2) ** 4
            g = np.array([-400.0 * p[0] * (p[1] - p[0] ** 2) - 2 * (1 - p[0]), 200.0 * (p[1] - p[0] ** 2), 6.0 * (p[2] - 0.3), 4 * (p[3] + 0.2) ** 3])
            return f, g
        bounds = ((-3., 0.8), (-3., 3.), (None, 0.25), (-0.1, None))
        ref = optimize.minimize(fg, np.array([-1.2, 1.0, 0.0, 0.5]), jac=True, method="L-BFGS-B", bounds=bounds, options={"disp": False})
        s = Lbfgsb(np.array([-1.2, 1.0, 0.0, 0.5]), bounds=bounds)
        while s.advance():
            s.feed(*fg(s.x.copy()))
        return bool(np.array_equal(s.x, ref.x) and float(s.f) == float(ref.fun) and s.nfev == ref.nfev and s.nit == ref.nit and s.status == ref.status)
    except Exception:      # noqa: BLE001
        return False
