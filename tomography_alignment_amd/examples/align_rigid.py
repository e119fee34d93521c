#!/usr/bin/env python3
"""
Joint reconstruction + rigid-body alignment -- the loop of the reference's examples/align_rigid.py:27-59:
alternate (a) SIRT with the current pose estimates and (b) a per-projection L-BFGS-B on cost_xzab / gradient_xzab
(tx, tz, alpha, beta; bounds +-3 px / +-0.02 rad).  Here (a) is the device-resident solver and (b) aligns all
projections in lock step, one fused cost/gradient launch per round (tomography_alignment_amd.alignment).

With a communicator (`comm`: tomography_alignment_amd.comm.RcclComm, one process per GPU) the projection angles are
sharded the way the reference shards them (np.array_split blocks, examples/mpi_reconstruct.py:34-38,
recon/sirt_mpi.py:40): rank r keeps ITS rows of the measured projections in HBM -- uploaded once, used by both halves
of every outer iteration -- reconstructs with the angle-sharded SIRT (recon/sirt_mpi.py: one sum of the voxel update
per iteration over xGMI) and aligns its own projections against the replicated reconstruction with no collective
inside the optimiser; the recovered pose table is summed once per outer iteration.  The reconstruction is warm-started
from the previous outer iteration's (examples/align_rigid.py:42) where it lies, in HBM.

    python -m tomography_alignment_amd.examples.align_rigid data.npz --outer 5 --sirt-iters 50
    python -m torch.distributed.run --nproc-per-node 8 -m tomography_alignment_amd.examples.align_rigid data.npz
"""
import argparse
import time

import numpy as np

from .. import alignment
from ..recon import sirt, sirt_mpi
from ..utilities import geometry

DEFAULT_BOUNDS = ((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02))        # examples/align_rigid.py:48


class OuterLoop(object):
    """The state of the alternation and its two halves as separate calls -- `run` below is `for it: reconstruct(); align()`.
    (Separate, so that a test can hand BOTH an unsharded and a sharded loop the same reconstruction / the same poses and compare
    each half at float32 accuracy: the composition itself is not comparable that tightly, see tests/test_dist_gloo.py.)

    data: dict with `projections` (n_proj, nx, nz) and `phi`; optional `phantom` (ground truth -> RMSE per outer iteration) and
    `xyz`, `alpha`, `beta` (true poses -> pose errors in the history).  `projections` may be a DEVICE buffer that already holds this
    rank's rows (then `projections_shape` = (n_proj, nx, nz) says what it is a block of), `phantom` a device buffer too.
    comm     None: one GPU.  A communicator: angle-sharded over its ranks (module docstring); every rank holds the same poses."""

    def __init__(self, data, backend=None, comm=None, kernel_names=None):
        self.comm = comm
        self.size = 1 if comm is None else (comm.Get_size() if hasattr(comm, "Get_size") else comm.size)
        self.rank = 0 if comm is None else (comm.Get_rank() if hasattr(comm, "Get_rank") else comm.rank)
        self.data = data
        self.phi = np.asarray(data["phi"], np.float64)
        n_proj = self.n_proj = self.phi.size
        proj = data["projections"]
        on_device = backend is not None and backend.is_buffer(proj)
        if on_device:
            _, nx, nz = (int(v) for v in data["projections_shape"])
        else:
            proj = np.asarray(proj, np.float32)
            nx, nz = proj.shape[1], proj.shape[2]
        ground_truth = data["phantom"] if "phantom" in data else None
        ny = int(data["ny"]) if "ny" in data else (ground_truth.shape[1] if getattr(ground_truth, "ndim", 0) == 3 else nx)
        self.vox_shape = (nx, ny, nz)
        self.geom = geometry.Geometry(n_proj, np.array([nx, ny, nz]), np.ones(3), np.array([nx, nz]), np.ones(2))
        self.mine = np.array_split(np.arange(n_proj), self.size)[self.rank]           # recon/sirt_mpi.py:40
        if backend is None:
            from ..backend import HipBackend
            backend = HipBackend(sirt_mpi.SIRT._shard_geometry(self.geom, self.mine), ctx=getattr(comm, "ctx", None))
        be = self.be = backend
        self.ctx = getattr(be, "ctx", None)
        self.kernel_names = kernel_names
        # this rank's measured rows and the ground truth go to HBM ONCE; every SIRT call and every alignment pass reads them there
        self.d_b = proj if on_device else be.upload(proj.reshape(n_proj, -1)[self.mine])
        if self.d_b.size != self.mine.size * nx * nz:
            raise ValueError("align_rigid: the device table of measured projections must hold this rank's %d rows" % self.mine.size)
        self.d_gt = None
        if ground_truth is not None:
            self.d_gt = ground_truth if be.is_buffer(ground_truth) else be.upload(np.asarray(ground_truth, np.float32).ravel())
        self.alpha_rec, self.beta_rec, self.xyz_rec = np.zeros(n_proj), np.zeros(n_proj), np.zeros((n_proj, 3))
        self.d_rec, self.solver = None, None

    def reconstruct(self, sirt_iters=50, positivity=True):
        """SIRT at the current pose estimates, warm-started from the previous reconstruction where it lies (examples/align_rigid.py:37-39,42).
        -> (iterations done, rms_error[:k]); the reconstruction is self.d_rec."""
        opts = {"_backend": self.be}
        if self.d_gt is not None:
            opts["ground_truth"] = self.d_gt
        if self.d_rec is not None:
            opts["rec"] = self.d_rec                                          # in place, in HBM
        angles = np.array([self.phi, self.alpha_rec, self.beta_rec]).T
        self.solver = None              # drop the previous solver's buffers BEFORE the next one allocates its own (d_rec is held separately)
        if self.comm is None:
            self.solver = sirt.SIRT(self.geom, self.d_b, angles, self.xyz_rec, options=opts)
        else:
            self.solver = sirt_mpi.SIRT(self.comm, self.geom, self.d_b, angles, self.xyz_rec, options=opts)
        k_done, err = self.solver.iterate_device(niter=sirt_iters, positivity=positivity)
        self.d_rec = self.solver.d_rec
        return k_done, err

    def align(self, bounds=DEFAULT_BOUNDS, **align_kwargs):
        """One alignment pass: every projection's L-BFGS-B on cost_xzab / gradient_xzab from a zero start against self.d_rec
        (examples/align_rigid.py:40-52); sets the pose estimates, returns alignment.align_projections' result dict."""
        kw = dict(letters="xzab", bounds=bounds)
        kw.update(align_kwargs)
        if self.comm is None:
            res = alignment.align_projections(self.be, self.d_rec, self.d_b, self.phi, indices=self.mine, **kw)
        else:
            res = alignment.align_projections_sharded(self.comm, self.be, self.d_rec, self.d_b, self.phi, **kw)
        self.xyz_rec = np.zeros((self.n_proj, 3))
        self.xyz_rec[:, 0], self.xyz_rec[:, 2] = res["x"][:, 0], res["x"][:, 1]
        self.alpha_rec, self.beta_rec = res["x"][:, 2].copy(), res["x"][:, 3].copy()
        return res

    def pose_errors(self):
        d = self.data
        if "xyz" not in d:
            return {}
        return {"shift_err_px": float(np.abs(self.xyz_rec[:, [0, 2]] - np.asarray(d["xyz"])[:, [0, 2]]).mean()),
                "tilt_err_deg": float(np.rad2deg(np.abs(np.column_stack([self.alpha_rec, self.beta_rec]) - np.column_stack([d["alpha"], d["beta"]])).mean()))}

    def download(self):
        return self.be.download(self.d_rec).reshape(self.vox_shape)


def run(data, n_outer=5, sirt_iters=50, bounds=DEFAULT_BOUNDS, verbose=True, backend=None, align_kwargs=None, comm=None,
        kernel_names=None, download=True, return_loop=False):
    """The loop of examples/align_rigid.py:36-52 (see OuterLoop for `data` and `comm`).
    kernel_names   with a HIP context: the history carries the HIP-event kernel time of each half of each outer iteration for these names.
    Returns (rec or None when not `download`, alpha, beta, xyz, history) -- with `return_loop` a sixth element, the OuterLoop itself
    (`loop.d_rec`: the reconstruction where it lies in HBM, `loop.solver`, `loop.d_b`).  Nothing is kept alive behind the caller's back:
    until round 5 the loop was parked in `run.last_loop`, which pinned about six volume-sized buffers until the next call."""
    loop = OuterLoop(data, backend=backend, comm=comm, kernel_names=kernel_names)
    ctx = loop.ctx
    history = []
    for it in range(n_outer):
        k0 = _kernel_ms(ctx, kernel_names)
        t0 = time.perf_counter()
        k_done, err = loop.reconstruct(sirt_iters)
        if ctx is not None:
            ctx.sync()
        t1 = time.perf_counter()
        k1 = _kernel_ms(ctx, kernel_names)
        res = loop.align(bounds, **(align_kwargs or {}))
        t2 = time.perf_counter()
        k2 = _kernel_ms(ctx, kernel_names)
        entry = {"outer": it, "rmse": float(err[-1]), "sirt_iterations": int(k_done), "residual": float(res["fun"].sum()), "launches": res["n_launch"],
                 "evals": res["n_eval"], "driver": res.get("driver"), "sirt_wall_s": round(t1 - t0, 3), "align_wall_s": round(t2 - t1, 3),
                 "ranks": loop.size}
        if kernel_names and ctx is not None:
            entry["sirt_kernel_ms"] = {k: round(k1[k] - k0[k], 1) for k in k1 if k1[k] - k0[k] > 0}
            entry["align_kernel_ms"] = {k: round(k2[k] - k1[k], 1) for k in k2 if k2[k] - k1[k] > 0}
        entry.update(loop.pose_errors())
        history.append(entry)
        if verbose and loop.rank == 0:
            print(entry)
    rec = loop.download() if (download and loop.d_rec is not None) else None
    out = (rec, loop.alpha_rec, loop.beta_rec, loop.xyz_rec, history)
    return out + (loop,) if return_loop else out


def _kernel_ms(ctx, names):
    if ctx is None or not names:
        return {}
    return {k: ctx.profile_get(k)[1] for k in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("data")
    ap.add_argument("--outer", type=int, default=5)
    ap.add_argument("--sirt-iters", type=int, default=50)
    a = ap.parse_args()
    import os
    comm = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:       # launched one process per GPU: python -m torch.distributed.run --nproc-per-node N -m ...
        from ..comm import RcclComm
        comm = RcclComm.from_env()
    run(dict(np.load(a.data)), a.outer, a.sirt_iters, comm=comm)
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
