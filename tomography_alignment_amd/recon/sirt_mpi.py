"""
Angle-sharded SIRT: the constructor / run_main_iteration signature of the reference's
recon/sirt_mpi.py:10-146 (`comm` first), with the mpi4py communicator replaced by RCCL over xGMI
(tomography_alignment_amd.comm.RcclComm) -- or any object exposing `size`, `rank`,
`allreduce_sum_(buffer)` and `allreduce_scalar(float)`.

Decomposition (recon/sirt_mpi.py:40-49): rank r owns the contiguous angle block
np.array_split(arange(n_proj), size)[r], its rows of b / W / residual, and a full replica of rec and V.
Per iteration: ONE all-reduce of the n_vox update V * A_r^T(W_r * res_r) (:101-103) plus one scalar
all-reduce of ||res_r||^2 (:110); every rank then applies the identical update, so no broadcast.
Reference quirks kept: zero guard `< 1e-8` (:69-70) and stop test `k > 1` (:116).
"""
import copy

import numpy as np

from .sirt import SIRT as _SIRT


class SIRT(_SIRT):

    def __init__(self, comm, geometry, projections, angles, xyz_shifts, options={}):
        self.comm = comm
        self.size = comm.Get_size() if hasattr(comm, "Get_size") else comm.size
        self.my_rank = comm.Get_rank() if hasattr(comm, "Get_rank") else comm.rank
        n_proj = angles.shape[0]
        self.my_index = np.array_split(np.arange(n_proj), self.size)[self.my_rank]     # sirt_mpi.py:40
        self.my_n_proj = np.size(self.my_index)
        opts = dict(options)
        if '_backend' not in opts and getattr(comm, "ctx", None) is not None:
            try:
                from ..backend import HipBackend
            except ImportError:
                from backend import HipBackend
            opts['_backend'] = HipBackend(self._shard_geometry(geometry, self.my_index), ctx=comm.ctx)
        super(SIRT, self).__init__(geometry, projections, angles, xyz_shifts, opts)

    @staticmethod
    def _shard_geometry(geometry, rows):
        g = copy.copy(geometry)                   # shallow: grids are shared, only the per-angle bookkeeping changes
        g.n_proj = int(np.size(rows))
        g.cor_shift = np.asarray(geometry.cor_shift)[rows].reshape(-1, 3)      # sirt_mpi.py:44-49
        return g

    def _my_rows(self):
        return self.my_index

    def _local_geometry(self, rows):
        return self._shard_geometry(self.geometry, rows)

    def _allreduce_vol(self, buf):
        return self.comm.allreduce_sum_(buf)

    def _allreduce_scalar(self, v):
        return self.comm.allreduce_scalar(v)

    def _is_root(self):
        return self.my_rank == 0

    n_pipeline_slabs = 8      # x slabs of the back-projection whose all-reduce overlaps the next slab's kernel (the last slab's is exposed)

    def _backproject_scaled(self):
        """recon/sirt_mpi.py:98-103 with the Allreduce pipelined: the volume is x-major, so the back-projection is done in
        a few x slabs (tile columns); as soon as a slab is final it is scaled by V and its all-reduce starts on the
        communication stream while the next slab is being back-projected.  Falls back to the plain sequence when the
        backend / communicator / poses do not offer the slab form."""
        be, comm = self.be, self.comm
        pipelined = (self.size > 1 or getattr(comm, "force_pipeline", False)) and hasattr(be, "adjoint_xslab") and \
            hasattr(comm, "allreduce_sum_async") and self.voxel_mask is None and self.n_pipeline_slabs > 1
        if pipelined:
            try:
                n_xt, tw = be.xslab_info()
                nx = int(self.geometry.vox_shape[0])
                plane = be.n_vox // nx
                cuts = np.unique(np.linspace(0, n_xt, min(self.n_pipeline_slabs, n_xt) + 1).astype(int))
                self.d_bp.zero_()
                for s in range(len(cuts) - 1):
                    be.adjoint_xslab(self.proj_mat.poses, self.d_res, self.d_bp, cuts[s], cuts[s + 1])
                    x_lo = 0 if s == 0 else min(nx, max(0, tw * int(cuts[s]) - 1))
                    x_hi = nx if s == len(cuts) - 2 else min(nx, max(0, tw * int(cuts[s + 1]) - 1))
                    if x_hi > x_lo:
                        seg_bp = self.d_bp.view(x_lo * plane, (x_hi - x_lo) * plane)
                        be.mul(seg_bp, self.d_V.view(x_lo * plane, (x_hi - x_lo) * plane))
                        comm.allreduce_sum_async(seg_bp)
                comm.join()
                return
            except Exception as e:                    # poses outside the tile kernels' domain: plain path from now on
                if "do not take the tile kernels" not in str(e):
                    raise
                self.n_pipeline_slabs = 1
        super(SIRT, self)._backproject_scaled()

    def _initialize(self):
        self._zero_guard = 1.e-8      # sirt_mpi.py:69-70
        self._stop_after = 1          # sirt_mpi.py:116
        super(SIRT, self)._initialize()

    def run_main_iteration(self, niter=100, positivity=False, make_plot=False, debug=False):
        return super(SIRT, self).run_main_iteration(niter=niter, make_plot=make_plot, positivity=positivity, debug=debug)
