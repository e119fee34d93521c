"""
Angle-sharded CGLS with the signature of the reference's recon/cgls_mpi.py:8-132 (`comm` first); RCCL
replaces the mpi4py Allreduce of A_r^T r_r (:55,98) and the scalar allreduces (:75-76,107).
The convergence monitor follows recon/cgls_mpi.py:74,80 literally: conv = ||b - A p|| summed over
ranks (the serial file monitors ||b - A rec||, recon/cgls.py:58 -- the two reference files differ).
"""
import numpy as np

from .cgls import CGLS as _CGLS
from .sirt_mpi import SIRT as _SIRTM

_SIRT_shard = _SIRTM._shard_geometry


class CGLS(_CGLS):

    def __init__(self, comm, geometry, projections, angles, xyz_shifts, options={}):
        self.comm = comm
        self.size = comm.Get_size() if hasattr(comm, "Get_size") else comm.size
        self.my_rank = comm.Get_rank() if hasattr(comm, "Get_rank") else comm.rank
        self.my_index = np.array_split(np.arange(angles.shape[0]), self.size)[self.my_rank]   # cgls_mpi.py:38
        self.my_n_proj = np.size(self.my_index)
        opts = dict(options)
        if '_backend' not in opts and getattr(comm, "ctx", None) is not None:
            try:
                from ..backend import HipBackend
            except ImportError:
                from backend import HipBackend
            opts['_backend'] = HipBackend(_SIRT_shard(geometry, self.my_index), ctx=comm.ctx)
        super(CGLS, self).__init__(geometry, projections, angles, xyz_shifts, opts)

    def _my_rows(self):
        return self.my_index

    def _local_geometry(self, rows):
        # recon/cgls_mpi.py:46 hands the FULL geometry to the projector, so rank r > 0 would read the first
        # my_n_proj rows of cor_shift; the rank's own rows are used here (as recon/sirt_mpi.py:44-49 does)
        return _SIRT_shard(self.geometry, rows)

    def _allreduce_vol(self, buf):
        return self.comm.allreduce_sum_(buf)

    def _allreduce_scalar(self, v):
        return self.comm.allreduce_scalar(v)

    def _conv_sumsq(self):
        return self.be.diff_sumsq(self.d_b, self.d_q)        # cgls_mpi.py:74

    def run_main_iteration(self, niter=100, make_plot=False):
        return super(CGLS, self).run_main_iteration(make_plot=make_plot, niter=niter)
