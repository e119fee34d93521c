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

    def _initialize(self):
        self._zero_guard = 1.e-8      # sirt_mpi.py:69-70
        self._stop_after = 1          # sirt_mpi.py:116
        super(SIRT, self)._initialize()

    def run_main_iteration(self, niter=100, positivity=False, make_plot=False, debug=False):
        return super(SIRT, self).run_main_iteration(niter=niter, make_plot=make_plot, positivity=positivity, debug=debug)
