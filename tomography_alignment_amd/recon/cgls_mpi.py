"""
Angle-sharded CGLS with the signature of the reference's recon/cgls_mpi.py:8-132 (`comm` first); RCCL
replaces the mpi4py Allreduce of A_r^T r_r (:55,98) and the scalar allreduces (:75-76,107).
The convergence monitor follows recon/cgls_mpi.py:74,80 literally: conv = ||b - A p|| summed over
ranks (the serial file monitors ||b - A rec||, recon/cgls.py:58 -- the two reference files differ).

Per iteration the reference does  A p -> two scalar allreduces -> A^T r -> Barrier + blocking Allreduce of the volume -> a third scalar
allreduce (:70-107).  The pipelined form here (round 5; decided collectively like the sharded SIRT's, recon/sirt_mpi.py::SlabPipeline):
    back-projection   slab by slab (tomo_adjoint_xslab); slab s's reduce-scatter runs on the communication stream while slab s + 1 is
                      back-projected; as the sums arrive, gamma = ||A^T r||^2 accumulates ON THE DEVICE over the rank's own pieces;
    scalars           ||A p||^2 and ||b - A p||^2 come from one pass over A p into two device accumulators and ONE small device-side
                      all-reduce; gamma and ||r||^2 likewise after the back-projection: two host synchronisations per iteration;
    direction update  p = s + beta p on the rank's own 1/P of each slab, all-gathered slab by slab while the NEXT iteration's A p of the
                      tile columns that are already complete is projected behind it (tomo_forward_xslab).
"""
import numpy as np

from .cgls import CGLS as _CGLS
from .sirt_mpi import SIRT as _SIRTM, SlabPipeline

_SIRT_shard = _SIRTM._shard_geometry


class CGLS(SlabPipeline, _CGLS):

    def __init__(self, comm, geometry, projections, angles, xyz_shifts, options={}):
        self.comm = comm
        self.size = comm.Get_size() if hasattr(comm, "Get_size") else comm.size
        self.my_rank = comm.Get_rank() if hasattr(comm, "Get_rank") else comm.rank
        self.my_index = np.array_split(np.arange(angles.shape[0]), self.size)[self.my_rank]   # cgls_mpi.py:38
        self.my_n_proj = np.size(self.my_index)
        self._pipelined = False
        self._q_ready = False
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

    def _initialize(self):
        decided = self.proj_mat is not None          # a re-initialisation (recon/cgls_mpi.py:84): the collective decision stands
        self._q_ready = False                        # whatever was projected ahead belonged to the direction being dropped
        super(CGLS, self)._initialize()              # plain sequence: A rec, b - A rec, blocking sum of A^T r (:50-56)
        if not decided:
            self._decide_pipeline(self.d_r, self.d_s)

    # ---- scalars: device accumulators, one host synchronisation per reduction point
    def _sum_accs(self, slot0, n):
        """The n device accumulators from slot0, summed over the ranks: one device-side all-reduce where the communicator is the
        backend's own RCCL one, else the local values through the communicator's small host all-reduce."""
        be, comm = self.be, self.comm
        if getattr(comm, "device_scalars", False) and getattr(be, "ctx", None) is getattr(comm, "ctx", None):
            return be.acc_fetch(slot0, n, allreduce=True)
        vals = np.array(be.acc_fetch(slot0, n), np.float64)
        if self.size > 1:
            comm.allreduce_array(vals)
        return vals

    def _apply_to_p(self):
        if self._q_ready:                            # projected slab by slab behind the previous iteration's all-gathers
            self._q_ready = False
            return
        super(CGLS, self)._apply_to_p()

    def _q_norms(self):
        """||A p||^2 and ||b - A p||^2 (cgls_mpi.py:72-76: two allreduces) from two accumulators and one collective."""
        be = self.be
        if not hasattr(be, "dot_acc"):
            return self._allreduce_scalar(be.dot(self.d_q, self.d_q)), self._allreduce_scalar(be.diff_sumsq(self.d_b, self.d_q))
        be.acc_zero(0, 2)
        be.dot_acc(self.d_q, self.d_q, 0)
        be.dot_acc(self.d_b, self.d_q, 1, diff=True)
        qq, dd = self._sum_accs(0, 2)
        return qq, dd

    def _own_chunk(self):
        """This rank's share of a volume-sized dot product in the plain sequence (every rank holds the whole, identical sum)."""
        n = self.be.n_vox
        piece = n // self.size
        return (self.my_rank * piece, piece), ((piece * self.size, n - piece * self.size) if self.my_rank == 0 else (0, 0))

    def _backproject_and_norms(self, want_rr):
        be, comm = self.be, self.comm
        self._iter_pipelined = self._pipe_now() and hasattr(be, "dot_acc")
        if not hasattr(be, "dot_acc"):
            self._allreduce_vol(self.proj_mat.T.apply(self.d_r, self.d_s))                   # cgls_mpi.py:95-98
            return be.dot(self.d_s, self.d_s), (self._allreduce_scalar(be.dot(self.d_r, self.d_r)) if want_rr else None)
        be.acc_zero(2, 2)
        if not self._iter_pipelined:
            self._allreduce_vol(self.proj_mat.T.apply(self.d_r, self.d_s))
            if want_rr:
                be.dot_acc(self.d_r, self.d_r, 3)
            for o, n in self._own_chunk():           # 1/P of the volume-sized sum per rank; the collective below completes it
                if n:
                    be.dot_acc(self.d_s.view(o, n), self.d_s.view(o, n), 2)
        else:
            plane, P, r = self._plane, self.size, self.my_rank
            self.d_s.zero_()
            for i, ((xt0, xt1), (x_lo, x_hi), _) in enumerate(self._plan):
                if self.my_n_proj > 0:
                    be.adjoint_xslab(self.proj_mat.poses, self.d_r, self.d_s, xt0, xt1, same_sinogram=(i > 0))
                o, n = x_lo * plane, (x_hi - x_lo) * plane
                piece, tail = self._pieces(n)
                # issued for EVERY slab on every rank (an empty one too): the same sequence of collectives everywhere
                if piece:
                    comm.reduce_scatter_sum_async(self.d_s.view(o, piece * P), piece)
                if tail:
                    comm.allreduce_sum_async(self.d_s.view(o + piece * P, tail))
            if want_rr:
                be.dot_acc(self.d_r, self.d_r, 3)    # on the compute stream while the last reductions are still on the links
            any_piece = any(self._pieces((x_hi - x_lo) * plane)[0] for _, (x_lo, x_hi), _ in self._plan)
            for _, (x_lo, x_hi), _ in self._plan:
                o, n = x_lo * plane, (x_hi - x_lo) * plane
                piece, tail = self._pieces(n)
                if piece:
                    comm.wait_next()                 # this slab's reduce-scatter: my piece of d_s is final
                    v = self.d_s.view(o + r * piece, piece)
                    be.dot_acc(v, v, 2)
                if tail:
                    comm.wait_next()
                    if r == 0 or not any_piece:      # identical on every rank: counted once (all-reduce form: every rank sums everything)
                        v = self.d_s.view(o + piece * P, tail)
                        be.dot_acc(v, v, 2)
            self._any_piece = any_piece
        pipelined_whole = self._iter_pipelined and not self._any_piece      # all-reduce form: gamma already complete on every rank
        if pipelined_whole:
            gamma = be.acc_fetch(2, 1)[0]
            rr = self._sum_accs(3, 1)[0] if want_rr else None
        else:
            gamma, rr = self._sum_accs(2, 2)
        return gamma, (rr if want_rr else None)

    def _update_p(self, beta, last=False):
        if not getattr(self, "_iter_pipelined", False):
            return super(CGLS, self)._update_p(beta, last)
        be, comm, plane, P, r = self.be, self.comm, self._plane, self.size, self.my_rank
        ahead = not last and self.my_n_proj > 0      # project A p for the next iteration (wasted only if a restart drops the direction)
        if ahead:
            self.d_q.zero_()                         # stream order: after `_r -= alpha q` has read it
        gathers, fwd_after = 0, None

        def forward_behind(cols):
            if ahead and cols is not None and cols[1] > cols[0]:
                be.forward_xslab(self.proj_mat.poses, self.d_p, self.d_q, cols[0], cols[1])

        for _, (x_lo, x_hi), cols in self._plan:
            o, n = x_lo * plane, (x_hi - x_lo) * plane
            piece, tail = self._pieces(n)
            if piece:
                be.xpay(self.d_p.view(o + r * piece, piece), self.d_s.view(o + r * piece, piece), beta)      # my 1/P of the slab
            if tail:
                be.xpay(self.d_p.view(o + piece * P, tail), self.d_s.view(o + piece * P, tail), beta)        # identical on every rank
            if piece:
                comm.allgather_async(self.d_p.view(o, piece * P), piece)
                gathers += 1
                if gathers > 1:                      # slab i's pieces travel while slab i + 1 is updated; the forward that needs slab i - 1 whole
                    comm.wait_next_gather()          # waits for ITS all-gather only
                    forward_behind(fwd_after)
                fwd_after = cols
            else:
                forward_behind(cols)                 # all-reduced slab: final on every rank as soon as it is updated
        if gathers:
            comm.wait_next_gather()
            forward_behind(fwd_after)
        comm.join()
        self._q_ready = ahead

    def _conv_sumsq(self):
        return self.be.diff_sumsq(self.d_b, self.d_q)        # cgls_mpi.py:74

    def run_main_iteration(self, niter=100, make_plot=False, _download=True):
        self._q_ready = False                        # a projection made ahead never outlives the call that made it
        return super(CGLS, self).run_main_iteration(make_plot=make_plot, niter=niter, _download=_download)
