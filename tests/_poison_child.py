"""Child process of tests/test_gpu_solvers.py::test_stream_overlap_waits_have_teeth.

Run as the FIRST AND ONLY context of a fresh process, so that the compute stream and the communication stream sit on hardware queues of
their own and really run side by side -- late in the pytest process (dozens of contexts opened before) the two streams were found on one
hardware queue, where the device itself serialises them and a missing wait can never show (gpurun_out/r4e_tests.log: the un-waited
control read saw restored values, and the control had to be demoted to "reported").  Here it is an assertion again:

  1. control        with the poison hook on (option comm_test_poison_us: the communication stream doubles a collective's buffer, idles,
                    halves it again, then runs the collective) a compute-stream read that did NOT wait sees the doubled buffer, one that
                    waited sees the original;
  2. solvers        the pipelined sharded SIRT and CGLS (reduce-scatter / own piece / all-gather per slab, the next forward projection
                    behind the all-gathers) with every collective poisoned equal the plain sequences;
  3. mutation       the same solvers through a communicator that DROPS one wait (one `wait_next()` or one `wait_next_gather()` in
                    recon/sirt_mpi.py / recon/cgls_mpi.py commented out, in effect) must NOT equal the plain sequence -- this is the
                    statement "the test fails when a wait is missing", executed on every run.

Prints one JSON line; exit code 0 only if all three hold."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


class DropOneWait(object):
    """An RcclComm whose n-th call of `which` does nothing."""

    def __init__(self, comm, which, nth):
        self._c, self._which, self._nth, self._seen = comm, which, nth, 0

    def __getattr__(self, name):
        attr = getattr(self._c, name)
        if name != self._which:
            return attr

        def maybe(*a, **k):
            self._seen += 1
            if self._seen == self._nth:
                return None
            return attr(*a, **k)
        return maybe


def main():
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.comm import RcclComm
    from tomography_alignment_amd.recon import sirt_mpi, cgls_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry
    from oracle import oracle as orc

    ctx = _lib.Context(0)                                     # the only context this process ever opens
    comm = RcclComm(ctx, 0, 1, RcclComm.unique_id(ctx.lib))
    out = {}
    # ---- 1. control
    geo0 = Geometry(2, np.array([16, 16, 16]), np.ones(3), np.array([16, 16]), np.ones(2))
    be0 = HipBackend(geo0, ctx=ctx)
    v = be0.upload(np.full(4096, 3.0, np.float32))
    comm.allreduce_sum_async(v)                               # warm-up: creates the communication stream, loads the RCCL kernels
    comm.wait_next()
    comm.join()
    ctx.sync()
    # The un-waited read must fall between the doubling kernel and the halving kernel of the poisoned collective.  No fixed sleep can promise
    # that on a loaded box: the idle time grows over up to four attempts (50 ms ... 1.6 s) and the host polls the compute stream's
    # view of the buffer during the first 40 % of it; one attempt that sees the doubled value is the control.  If none does (the two streams
    # share a hardware queue, or the box stalls for seconds), the control reports "could not race" -- NOT a failure of the solvers: the
    # mutation cases below (a dropped wait changes the result) are the hard evidence that an un-waited read can be caught here.
    out["control_attempts"] = []
    raced = False
    for poison_us in (50000, 200000, 800000, 1600000):
        v.upload(np.full(4096, 3.0, np.float32))
        ctx.sync()
        ctx.set_option("comm_test_poison_us", poison_us)
        t0 = time.perf_counter()
        comm.allreduce_sum_async(v)
        seen = None
        while time.perf_counter() - t0 < 0.4e-6 * poison_us:
            seen = be0.dot(v, v)                              # NOT waited for (be.dot synchronises the compute stream only)
            if seen == 4096 * 36.0:
                break
            time.sleep(0.002)
        comm.wait_next()
        waited = be0.dot(v, v)
        comm.join()
        ctx.set_option("comm_test_poison_us", 0)
        out["control_attempts"].append([poison_us, seen, waited])
        out["unwaited"], out["waited"] = seen, waited
        if seen == 4096 * 36.0 and waited == 4096 * 9.0:
            raced = True
            break
        if waited != 4096 * 9.0:
            break                                             # a WAITED read that is wrong is a real failure: stop and report it
    out["control_raced"] = raced
    out["control_ok"] = bool(out["waited"] == 4096 * 9.0)     # hard: after the wait the buffer is whole again

    # ---- 2. + 3. the solvers: ragged volume, 6 tile columns, flat and tilted poses, positivity, ground truth
    shape, ndet, n_proj = (80, 40, 200), (72, 210), 12
    rng = np.random.default_rng(11)
    x = np.zeros(shape, np.float32)
    x[10:70, 6:34, 70:150] = rng.uniform(0.2, 1.0, (60, 28, 80)).astype(np.float32)
    phi = np.linspace(0.05, np.pi - 0.05, n_proj)
    geo = Geometry(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2))
    og = orc.Geo(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2))
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b)) / np.max(np.abs(b)))      # noqa: E731
    worst_ok, least_broken, cases = 0.0, np.inf, []
    for tilt in (0.0, 1.0):
        alpha, beta = np.deg2rad(tilt * rng.uniform(-1.5, 1.5, n_proj)), np.deg2rad(tilt * rng.uniform(-1.5, 1.5, n_proj))
        xyz = np.zeros((n_proj, 3))
        xyz[:, 0], xyz[:, 2] = rng.uniform(-2, 2, n_proj), rng.uniform(-2, 2, n_proj)
        b = orc.forward(og, x, alpha=alpha, beta=beta, phi=phi, xyz_shift=xyz).astype(np.float32)
        ang = np.array([phi, alpha, beta]).T

        def solve(kind, c, force, poison, slabs=6, shard=True, niter=4):
            c.force_pipeline = force
            opts = {"_backend": HipBackend(geo, ctx=ctx)}
            if kind == "sirt":
                opts["ground_truth"] = x
                s = sirt_mpi.SIRT(c, geo, b.copy(), ang, xyz, options=opts)
            else:
                s = cgls_mpi.CGLS(c, geo, b.copy(), ang, xyz, options=opts)
            s.n_pipeline_slabs, s.shard_update = slabs, shard
            ctx.set_option("comm_test_poison_us", poison)
            try:
                r = s.run_main_iteration(niter=niter, positivity=True) if kind == "sirt" else s.run_main_iteration(niter=niter)
            finally:
                ctx.set_option("comm_test_poison_us", 0)
                comm.join()
            assert s._pipelined == force
            return np.asarray(r[0], np.float64).ravel(), r[1]

        for kind in ("sirt", "cgls"):
            ref, ref_err = solve(kind, comm, False, 0)
            for slabs, shard in ((6, True), (6, False), (3, True)):
                got, err = solve(kind, comm, True, 300, slabs, shard)
                d = rel(got, ref)
                worst_ok = max(worst_ok, d)
                cases.append((kind, tilt, slabs, shard, "poisoned", d))
            # a wait goes missing: the 1st / 2nd reduction wait, the 2nd all-gather wait.  (Not every wait can show on ONE communication
            # stream: in SIRT's update the all-gather of slab 1 is queued behind every reduce-scatter of the iteration, so once the
            # compute stream has waited for an all-gather, all reductions are complete and the reduction waits after it are implied --
            # dropping SIRT's 5th `wait_next` changes nothing, measured; the waits below precede any such implication.)
            for which, nth in (("wait_next", 1), ("wait_next", 2), ("wait_next_gather", 2)):
                got, err = solve(kind, DropOneWait(comm, which, nth), True, 2000)
                d = rel(np.nan_to_num(got, nan=1e30, posinf=1e30, neginf=-1e30), ref)
                least_broken = min(least_broken, d)
                cases.append((kind, tilt, which, nth, "wait dropped", d))
    out["poisoned_worst_rel"] = worst_ok
    out["wait_dropped_least_rel"] = float(least_broken)
    out["cases"] = cases
    out["solvers_ok"] = bool(worst_ok < 2e-6)
    out["mutation_ok"] = bool(least_broken > 1e-3)
    comm.close()
    print(json.dumps(out))
    return 0 if (out["control_ok"] and out["solvers_ok"] and out["mutation_ok"]) else 1


if __name__ == "__main__":
    sys.exit(main())
