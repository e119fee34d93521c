"""
Batched per-projection pose alignment: the inner loop of the reference's examples/align_rigid.py:40-52
(for every projection: scipy L-BFGS-B on cost_xzab / gradient_xzab with bounds) with all projections'
optimisers advancing together, so that each round of function evaluations is ONE launch of the fused
cost/gradient kernel over many projections (tomo_cost_grad) instead of n_proj separate launches.

scipy's L-BFGS-B is kept as the optimiser (drop-in semantics, identical iterates).  Round 4: its compiled core is driven in reverse
communication for all projections from ONE Python thread (_lbfgsb_batch.py): the live optimisers are two populations that take
turns -- while the GPU evaluates one (a helper thread sits in the ctypes call, the GIL released), this thread runs the other's
optimiser steps; the few optimisers of the tail run as one population.  Where this scipy's private core does not have the
expected layout (or `driver="threads"`), round 3's form runs instead: a fixed pool of worker threads each runs one projection's
`optimize.minimize` at a time; an evaluation request blocks until the scheduler has collected the pending requests of the live
workers, evaluated them in one batch and handed the results back (a third of the wall time went to thread hand-overs).

Multi-GPU: projections are independent -- rank r aligns np.array_split(arange(n_proj), P)[r] with a
replicated volume and no collective inside the optimiser; the 4 recovered parameters per projection are gathered
once at the end (an all-reduce of a zero-padded table).
"""
import sys
import threading
import time

import numpy as np
from scipy import optimize

try:
    from . import _lib, _lbfgsb_batch
except ImportError:
    import _lib
    import _lbfgsb_batch

_BATCH_OK = None


def batch_driver_available():
    """The reverse-communication driver reproduces scipy.optimize.minimize bit for bit on this scipy (checked once)?"""
    global _BATCH_OK
    if _BATCH_OK is None:
        _BATCH_OK = bool(_lbfgsb_batch.AVAILABLE and _lbfgsb_batch.self_test())
    return _BATCH_OK

# parameter letters -> pose columns (phi, alpha, beta, tx, ty, tz, cor_x) and Jacobian rows (tx,ty,tz,phi,alpha,beta)
_POSE_COL = {"x": 3, "y": 4, "z": 5, "p": 0, "a": 1, "b": 2}
_GRAD_ROW = {"x": 0, "y": 1, "z": 2, "p": 3, "a": 4, "b": 5}


class BatchEvaluator(object):
    """cost / gradient of many projections per launch, volume and measured projections resident in HBM.

    Only the measured rows this evaluator will ever be asked about go to the device (`indices`: a rank of a sharded pass uploads its own
    np.array_split block, not all n_proj rows -- 4.3 GB per rank at 1024^2 x 1024 otherwise); `projections` may also BE a device table
    already holding exactly those rows, in the order of `indices` (the sharded solver's own rows: nothing crosses PCIe)."""

    def __init__(self, backend, rec, projections, cor_shift=None, trace=None, indices=None, n_all=None):
        self.be = backend
        self.trace = trace      # a list: every launch's (table rows, poses) is appended (bench.py replays them in full batches)
        self.vol = rec if backend.is_buffer(rec) else backend.upload(np.asarray(rec, np.float32).ravel())
        if backend.is_buffer(projections):
            if indices is None or n_all is None:
                raise ValueError("BatchEvaluator: a device table of measured rows needs `indices` (the projections its rows belong to) and `n_all`")
            indices = np.asarray(indices, np.int64)
            if projections.size != indices.size * backend.n_det:
                raise ValueError("BatchEvaluator: device table has %d values for %d rows of %d" % (projections.size, indices.size, backend.n_det))
            self.n = int(n_all)
            self._b_all = projections
        else:
            b = np.asarray(projections, np.float32)
            self.n = b.shape[0]
            indices = np.arange(self.n) if indices is None else np.asarray(indices, np.int64)
            b = b.reshape(self.n, -1)
            self._b_all = backend.upload(b if indices.size == self.n and np.array_equal(indices, np.arange(self.n)) else b[indices])
        self.row_of = np.full(self.n, -1, np.int64)      # projection index -> row of the device table (-1: not resident)
        self.row_of[indices] = np.arange(indices.size)
        self.cor = np.zeros(self.n) if cor_shift is None else np.asarray(cor_shift, np.float64).reshape(self.n, -1)[:, 0]
        self._staged = False
        self.n_launch = 0
        self.n_eval = 0
        self.t_eval = 0.0       # seconds inside the evaluation calls (kernel + staging), for reports

    def evaluate(self, idx, poses6):
        """idx: projection indices (m,), poses6: (m,6) rows (phi, alpha, beta, tx, ty, tz) -> cost[m], grad6[m,6]."""
        idx = np.asarray(idx, np.int64)
        m = idx.size
        rows = self.row_of[idx]
        if m and rows.min() < 0:
            raise ValueError("BatchEvaluator: projection %d is not among the rows this evaluator holds" % int(idx[np.argmin(rows)]))
        poses = np.zeros((m, _lib.POSE_STRIDE), np.float64)
        poses[:, :6] = poses6
        poses[:, 6] = self.cor[idx]
        self.n_launch += 1
        self.n_eval += m
        if self.trace is not None:
            self.trace.append((rows.copy(), poses.copy()))
        t0 = time.perf_counter()
        out = self.be.cost_grad(np.ascontiguousarray(poses), self.vol, self._b_all, rows=rows)   # measured rows stay in HBM
        if not self._staged:                    # self.vol is pinned until close(): its zero-padded copy is staged once
            self._staged = True
            ctx = getattr(self.be, "ctx", None)
            if ctx is not None:
                ctx.set_option("reuse_staged_volume", 1)
        self.t_eval += time.perf_counter() - t0
        return out

    def close(self):
        """Give the context back: the next caller's volume is staged afresh."""
        ctx = getattr(self.be, "ctx", None)
        if self._staged and ctx is not None:
            ctx.set_option("reuse_staged_volume", 0)
        self._staged = False


class _Scheduler(object):
    """Batches the workers' evaluation requests.  A launch goes out as soon as at least half of the live workers are
    waiting (all of them once few are left), so the Python side of scipy's iterations for one half overlaps the kernel
    of the other half (ctypes drops the GIL during the call).  Every worker sleeps on its own Event: a request wakes
    the scheduler only, a finished batch wakes exactly its workers."""

    def __init__(self, evaluator, small=8):
        self.ev = evaluator
        self.lock = threading.Lock()
        self.wake = threading.Condition(self.lock)        # the scheduler alone waits here
        self.live = 0
        self.pending = {}
        self.results = {}
        self.events = {}
        self.error = None
        self.small = small

    def enter(self, i):
        with self.lock:
            self.live += 1
            self.events[i] = threading.Event()

    def leave(self, i):
        with self.lock:
            self.live -= 1
            del self.events[i]
            self.wake.notify()

    def swap(self, i, j):
        """Worker finished projection i and starts projection j: the number of live optimisers does not change."""
        with self.lock:
            del self.events[i]
            self.events[j] = threading.Event()

    def _ready(self):
        n = len(self.pending)
        return n > 0 and (n == self.live or (self.live > self.small and 2 * n >= self.live))

    def request(self, i, pose6):
        ev = self.events[i]
        with self.lock:
            if self.error is not None:
                raise self.error
            self.pending[i] = pose6
            if self._ready():
                self.wake.notify()
        ev.wait()
        ev.clear()
        with self.lock:
            if self.error is not None:
                raise self.error
            return self.results.pop(i)

    def run(self, finished):
        """Serve requests until `finished()` (called under the lock) says no worker is left or to come."""
        while True:
            with self.lock:
                while not self._ready():
                    if finished() and self.live == 0:
                        return
                    self.wake.wait(0.05)
                idx = sorted(self.pending)
                poses = np.array([self.pending[i] for i in idx])
                self.pending.clear()
            try:
                cost, g6 = self.ev.evaluate(idx, poses)
            except Exception as e:                      # release the workers, re-raise in the caller
                with self.lock:
                    self.error = e
                    for ev in self.events.values():
                        ev.set()
                raise
            with self.lock:
                for k, i in enumerate(idx):
                    self.results[i] = (float(cost[k]), g6[k].copy())
                    self.events[i].set()


_WARNED_FALLBACK = False


def align_projections(backend, rec, projections, phi, letters="xzab", x0=None, angles0=None, xyz0=None, cor_shift=None,
                      bounds=None, scale_factor=None, options=None, indices=None, max_threads=256, driver="auto", trace=None):
    """Align many projections at once.

    letters   which pose components are free, reference naming (utilities/alignment_functions.py): "xzab" = tx, tz,
              alpha, beta (the pair examples/align_rigid.py:46-49 minimises), "xzpab", "xz", ...
    x0        (n, len(letters)) start values (default 0); angles0 (n,3) fixed (phi,alpha,beta) offsets (default
              phi from `phi`, 0, 0); xyz0 (n,3) fixed translations.
    bounds    per-parameter (lo, hi) pairs as scipy takes them (examples/align_rigid.py:48 uses +-3 px / +-0.02 rad).
    projections   host array of ALL measured projections (n, ...) -- only the rows `indices` names are uploaded -- or a device table
              that already holds exactly the rows of `indices`, in that order.
    indices   the projections to align (default all); the other rows of the returned tables stay zero.
    Returns dict(x=(n,k), fun=(n,), nfev=(n,), n_launch, n_eval).
    """
    n_all = int(np.size(phi))
    if not backend.is_buffer(projections):
        projections = np.asarray(projections, np.float32)
        if projections.shape[0] != n_all:
            raise ValueError("align_projections: %d projections for %d angles" % (projections.shape[0], n_all))
    indices = np.arange(n_all) if indices is None else np.asarray(indices, np.int64)
    k = len(letters)
    cols = [_POSE_COL[c] for c in letters]
    rows = [_GRAD_ROW[c] for c in letters]
    scale = np.ones(k) if scale_factor is None else np.asarray(scale_factor, np.float64)
    base = np.zeros((n_all, 6))
    base[:, 0] = np.asarray(phi, np.float64)
    if angles0 is not None:
        base[:, 0:3] = np.asarray(angles0, np.float64)
    if xyz0 is not None:
        base[:, 3:6] = np.asarray(xyz0, np.float64)
    x0 = np.zeros((n_all, k)) if x0 is None else np.asarray(x0, np.float64).reshape(n_all, k)
    ev = BatchEvaluator(backend, rec, projections, cor_shift, trace=trace, indices=indices, n_all=n_all)
    out_x, out_f, out_n = np.zeros((n_all, k)), np.zeros(n_all), np.zeros(n_all, np.int64)
    opts = {"disp": False}
    opts.update(options or {})

    if driver == "batch" and not batch_driver_available():
        raise _lib.TomoError("align_projections: the reverse-communication L-BFGS-B driver does not match this scipy")
    if driver != "threads" and batch_driver_available():
        # ---- round 4: every optimiser in this thread, two populations taking turns on the device (see the module docstring)
        from concurrent.futures import ThreadPoolExecutor
        order = np.asarray(indices, np.int64)

        def fun_batch(ids, X):
            sel = order[np.asarray(ids, np.int64)]
            poses = base[sel].copy()
            poses[:, cols] += X
            cost, g6 = ev.evaluate(sel, poses)
            return cost, g6[:, rows] * scale

        t0 = time.perf_counter()
        ctx = getattr(backend, "ctx", None)
        # the helper thread is the only one that touches the GPU while the pass runs -- minimize_many sends EVERY evaluation through
        # `overlap`, the tail and small problem counts included; HIP's current device is per thread, so that thread binds it once
        with ThreadPoolExecutor(max_workers=1, initializer=(ctx.make_current if ctx is not None else None)) as pool:
            try:
                x, f, nf, _ = _lbfgsb_batch.minimize_many(fun_batch, x0[order], bounds=bounds, options=opts, overlap=pool.submit)
            finally:
                ev.close()
        out_x[order], out_f[order], out_n[order] = x, f, nf
        return {"x": out_x, "fun": out_f, "nfev": out_n, "n_launch": ev.n_launch, "n_eval": ev.n_eval, "t_eval": ev.t_eval,
                "driver": "batch", "wall_s": time.perf_counter() - t0}

    global _WARNED_FALLBACK
    if driver != "threads" and not _WARNED_FALLBACK:
        _WARNED_FALLBACK = True
        import warnings
        warnings.warn("align_projections: this scipy's private L-BFGS-B core does not have the layout _lbfgsb_batch.py was written against "
                      "(scipy 1.15.x); falling back to one optimize.minimize per worker thread (same results, slower host side)", RuntimeWarning)
    # a fixed pool of worker threads (at most max_threads): each runs one projection's optimiser at a time and takes the next
    # projection from the queue when it converges, so the batches stay full until the very end instead of draining once per
    # chunk -- and 720 projections cost 256 threads, not 720
    import collections
    sched = _Scheduler(ev)
    errors = []
    order = [int(i) for i in indices]
    n_workers = min(max_threads, len(order))
    queue = collections.deque(order[n_workers:])
    qlock = threading.Lock()

    def solve(i):
        def fun(p):
            pose = base[i].copy()
            pose[cols] += p
            cost, g6 = sched.request(i, pose)
            return cost, g6[rows] * scale
        res = optimize.minimize(fun, x0[i], jac=True, method="L-BFGS-B", bounds=bounds, options=opts)
        out_x[i], out_f[i], out_n[i] = res.x, res.fun, res.nfev

    def work(i):
        try:
            while True:
                solve(i)
                with qlock:
                    nxt = queue.popleft() if (queue and not errors) else None
                if nxt is None:
                    break
                sched.swap(i, nxt)
                i = nxt
        except Exception as e:      # noqa: BLE001
            errors.append(e)
        finally:
            sched.leave(i)

    for i in order[:n_workers]:                        # the first pool is registered as a whole, so that the first launch waits
        sched.enter(i)                                 # for all of it instead of firing on the first arrival
    threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in order[:n_workers]]
    switch = sys.getswitchinterval()
    sys.setswitchinterval(min(switch, 5e-4))           # hundreds of short-running threads: hand the GIL over promptly
    for t in threads:
        t.start()
    try:
        sched.run(lambda: True)                        # returns once no worker is live (the queue is drained by then)
    finally:
        for t in threads:
            t.join(timeout=60)
        ev.close()
        sys.setswitchinterval(switch)
    if errors:
        raise errors[0]
    return {"x": out_x, "fun": out_f, "nfev": out_n, "n_launch": ev.n_launch, "n_eval": ev.n_eval, "t_eval": ev.t_eval, "driver": "threads"}


def align_projections_sharded(comm, backend, rec, projections, phi, **kw):
    """Rank r aligns its np.array_split block of the projections; every rank returns the full (n, k) table.
    `projections`: the host array of all measured projections (only the rank's own rows are uploaded) or a device table holding
    exactly the rank's own rows (the sharded solver's `d_b`)."""
    n = int(np.size(phi))
    size = comm.Get_size() if hasattr(comm, "Get_size") else comm.size
    rank = comm.Get_rank() if hasattr(comm, "Get_rank") else comm.rank
    mine = np.array_split(np.arange(n), size)[rank]
    if mine.size:
        res = align_projections(backend, rec, projections, phi, indices=mine, **kw)
    else:       # more ranks than projections: nothing to align here, but the table all-reduce below is issued by every rank
        k0 = len(kw.get("letters", "xzab"))
        res = {"x": np.zeros((n, k0)), "fun": np.zeros(n), "nfev": np.zeros(n, np.int64), "n_launch": 0, "n_eval": 0}
    k = res["x"].shape[1]
    table = np.zeros((n, k + 2))
    table[mine, :k] = res["x"][mine]
    table[mine, k] = res["fun"][mine]
    table[mine, k + 1] = res["nfev"][mine]
    if size > 1:
        comm.allreduce_array(table)                      # tiny: k + 2 numbers per projection, zero outside the own block
    out = dict(res)
    out.update({"x": table[:, :k], "fun": table[:, k], "nfev": table[:, k + 1].astype(np.int64)})
    return out
