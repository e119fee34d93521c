"""
Angle-sharded SIRT: the constructor / run_main_iteration signature of the reference's
recon/sirt_mpi.py:10-146 (`comm` first), with the mpi4py communicator replaced by RCCL over xGMI
(tomography_alignment_amd.comm.RcclComm) -- or any object exposing `size`, `rank`,
`allreduce_sum_(buffer)` and `allreduce_scalar(float)`.

Decomposition (recon/sirt_mpi.py:40-49): rank r owns the contiguous angle block
np.array_split(arange(n_proj), size)[r], its rows of b / W / residual, and a full replica of rec and V.
Per iteration: ONE sum over the ranks of the n_vox update V * A_r^T(W_r * res_r) (:101-103) plus one scalar
all-reduce of ||res_r||^2 (:110).  The reference all-reduces and every rank then applies the identical update
to its replica; the pipelined form here (round 4) reduce-scatters instead: a rank receives the sum of ITS 1/P of
each slab, updates that piece of `rec` (V scaling, positivity, error sum: 1/P of the volume-sized vector work)
and the pieces are all-gathered -- the same bytes on the links (a ring all-reduce is these two phases).
Reference quirks kept: zero guard `< 1e-8` (:69-70) and stop test `k > 1` (:116).
"""
import copy

import numpy as np

from .sirt import SIRT as _SIRT


class SlabPipeline(object):
    """What the pipelined iterations of the angle-sharded solvers share (SIRT below, recon/cgls_mpi.py): the cut of the volume into x slabs
    = ranges of tile columns, the ONE collective decision whether every rank can take the slab calls, and the split of a slab into
    one piece per rank (+ a tail).  Expects self.be, self.comm, self.size, self.my_n_proj, self.geometry, self.proj_mat,
    self._allreduce_scalar and (optionally) self.voxel_mask."""

    n_pipeline_slabs = 8      # x slabs per iteration: slab s's collective overlaps the back-projection of the later slabs and the
                              # update + next forward projection of the earlier ones
    shard_update = True       # reduce-scatter -> vector work on the rank's own 1/P of each slab -> all-gather (False: all-reduce + the
                              # identical vector work on every rank, the round-3 form; same on every rank)

    def _slab_plan(self):
        """[(tile columns to back-project, voxel x range they finalise, tile columns whose forward may start after it)]."""
        n_xt, tw = self.be.xslab_info()
        nx = int(self.geometry.vox_shape[0])
        cuts = np.linspace(0, n_xt, min(self.n_pipeline_slabs, n_xt) + 1)
        if n_xt >= 8 * self.n_pipeline_slabs:
            # whole multiples of 4 tile columns (= 8 of the gather back-projection's 8-voxel column tiles, one row of its XCD patches):
            # a slab whose patch grid is ragged or too small to use runs 10 % slower (profiles/round3_sharded_slabs.md)
            cuts = np.round(cuts / 4.0) * 4.0
            cuts[-1] = n_xt
        cuts = np.unique(cuts.astype(int))
        plan, f_done = [], 0
        for s in range(len(cuts) - 1):
            last = s == len(cuts) - 2
            x_lo = 0 if s == 0 else min(nx, max(0, tw * int(cuts[s]) - 1))
            x_hi = nx if last else min(nx, max(0, tw * int(cuts[s + 1]) - 1))
            # tile column t reads the voxels x in [tw*t - 1, tw*t + tw]: final once every x < x_hi is, i.e. t <= cuts[s+1] - 2
            f_end = n_xt if last else max(f_done, int(cuts[s + 1]) - 1)
            plan.append(((int(cuts[s]), int(cuts[s + 1])), (x_lo, x_hi), (f_done, f_end)))
            f_done = f_end
        return plan, nx

    def _decide_pipeline(self, probe_proj, probe_vol):
        """Rank-uniform by construction: one scalar all-reduce that EVERY rank issues, whatever it found locally.
        probe_proj / probe_vol: a sinogram-sized and a volume-sized buffer of the solver for the (empty) probing slab calls."""
        be, comm = self.be, self.comm
        able = all(hasattr(be, a) for a in ("adjoint_xslab", "forward_xslab", "xslab_info", "tiles_take", "update_acc")) and \
            all(hasattr(comm, a) for a in ("allreduce_sum_async", "wait_next", "join")) and getattr(self, "voxel_mask", None) is None
        if able and self.my_n_proj > 0:
            able = bool(be.tiles_take(self.proj_mat.poses, probe_proj, probe_vol))
        n_unable = self._allreduce_scalar(0.0 if able else 1.0)
        # what every rank CAN do is settled here, collectively; whether the caller WANTS slabs (n_pipeline_slabs, the same on every rank)
        # is read when an iteration starts (_pipe_now), so it may be set after construction
        self._pipelined = bool((self.size > 1 or getattr(comm, "force_pipeline", False)) and n_unable == 0)
        self._plan_slabs = None

    def _pipe_now(self):
        """Pipelined form for the iteration that starts now?  (n_pipeline_slabs may be changed between iterations -- on every
        rank alike; <= 1 means the plain sequence.)"""
        if not self._pipelined or self.n_pipeline_slabs <= 1:
            return False
        if self._plan_slabs != self.n_pipeline_slabs:
            self._plan, nx = self._slab_plan()
            self._plane = self.be.n_vox // nx
            self._plan_slabs = self.n_pipeline_slabs
        return True

    def _pieces(self, n):
        """A slab of n voxels as size equal pieces + a tail of < size voxels (all-reduced and worked on by every rank)."""
        piece = n // self.size if self.shard_update and all(hasattr(self.comm, a) for a in ("reduce_scatter_sum_async", "allgather_async", "wait_next_gather")) else 0
        return piece, n - piece * self.size


class SIRT(SlabPipeline, _SIRT):

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

    def _allreduce_host(self, a):
        """Sum of a host array over the ranks (run_regularized_gradient_descent: recon/sirt_mpi.py:160-178)."""
        if self.size == 1:
            return a
        a = np.asarray(a)
        if a.dtype == np.float32 and a.size > 4096:
            # volume-sized float32 (the gradient of run_regularized_gradient_descent in the default precision): the float32 DEVICE collective
            buf = self.be.upload(a.ravel())
            self.comm.allreduce_sum_(buf)
            return self.be.download(buf).reshape(a.shape)
        # everything else keeps float64 end to end, as the reference's MPI.DOUBLE Allreduce does for precision=np.float64
        # (recon/sirt_mpi.py:159-162): allreduce_array is a float64 collective through a pinned staging buffer that only ever grows, so a
        # large array goes through it in pieces of 1 M values (the arithmetic must not change with the array's size)
        out = np.array(a, np.float64).ravel()
        for lo in range(0, out.size, 1 << 20):
            self.comm.allreduce_array(out[lo:lo + (1 << 20)])
        return out.reshape(a.shape).astype(a.dtype, copy=False)

    def _is_root(self):
        return self.my_rank == 0

    # ---- the pipelined iteration --------------------------------------------------------------------------------------------
    # The reference does, per iteration, forward -> residual -> back-projection -> ONE blocking Allreduce of the whole volume ->
    # update (recon/sirt_mpi.py:92-110).  Here the volume (x-major) is cut into a few x slabs = ranges of tile columns:
    #     back-projection:  slab s is back-projected (tomo_adjoint_xslab), scaled by V, and its all-reduce starts on the
    #                       communication stream while slab s + 1 is being back-projected;
    #     update + forward: as soon as slab s's all-reduce is done (tomo_comm_wait_next) slab s of `rec` is updated and the NEXT
    #                       iteration's forward projection of the tile columns that read only final voxels starts
    #                       (tomo_forward_xslab) -- while the all-reduces of the later slabs are still on the links.
    # Only the first slab's all-reduce has nothing but back-projection to hide behind, and only the last slab's update + forward
    # has no communication beside it.
    # WHETHER the pipelined form is used is decided ONCE, COLLECTIVELY, in _initialize: every rank reports whether its own angle
    # block takes the tile kernels (and its backend / communicator offer the slab calls); all ranks pipeline or none does.  (Until
    # round 3 each rank decided from its own block inside the loop: a rank holding a pose the tile kernels decline fell back to one
    # whole-volume all-reduce while its peers issued eight slab all-reduces -- mismatched collectives, VERDICT r2 #13.)
    def _forward(self):
        if self._pipelined and self._ax_ready:      # already projected slab by slab behind the previous iteration's update
            self._ax_ready = False
            return
        super(SIRT, self)._forward()

    def _backproject_scaled(self):
        """recon/sirt_mpi.py:98-103; pipelined form: see above (the reductions are consumed by _update)."""
        self._iter_pipelined = self._pipe_now()
        if not self._iter_pipelined:
            return super(SIRT, self)._backproject_scaled()
        be, comm, plane = self.be, self.comm, self._plane
        self.d_bp.zero_()
        for i, ((xt0, xt1), (x_lo, x_hi), _) in enumerate(self._plan):
            if self.my_n_proj > 0:
                be.adjoint_xslab(self.proj_mat.poses, self.d_res, self.d_bp, xt0, xt1, same_sinogram=(i > 0))
            o, n = x_lo * plane, (x_hi - x_lo) * plane
            piece, tail = self._pieces(n)
            # issued for EVERY slab on every rank (an empty one too): same sequence of collectives everywhere
            # (the scaling by V -- sirt_mpi.py:101 -- commutes with the sum: applied by the update)
            if piece:
                comm.reduce_scatter_sum_async(self.d_bp.view(o, piece * self.size), piece)
            if tail:
                comm.allreduce_sum_async(self.d_bp.view(o + piece * self.size, tail))

    def _update(self, positivity, last=False):
        if not self._iter_pipelined:
            return super(SIRT, self)._update(positivity, last)
        be, comm, plane, P, r = self.be, self.comm, self._plane, self.size, self.my_rank
        ahead = not last and self.my_n_proj > 0     # project for the next iteration (wasted only if the stop rule fires now)
        if ahead:
            self.d_ax.zero_()                       # stream order: after the residual kernel has read it
        first, gathers, fwd_after = True, 0, None
        # with pieces, a rank sums (gt - rec)^2 over its own voxels, rank 0 adds the tails, and one scalar all-reduce completes it; in the
        # all-reduce form (no pieces anywhere) every rank updates -- and sums over -- everything, as in round 3, and nothing is reduced
        any_piece = any(self._pieces((x_hi - x_lo) * plane)[0] for _, (x_lo, x_hi), _ in self._plan)

        def upd(o, n, with_gt):
            nonlocal first
            be.update_acc(self.d_rec.view(o, n), self.d_bp.view(o, n), self.d_V.view(o, n), positivity,
                          self.d_gt.view(o, n) if (self.d_gt is not None and with_gt) else None, first=first)
            first = False

        def forward_behind(cols):
            """The next iteration's forward projection of the tile columns a finished slab completes."""
            if ahead and cols is not None and cols[1] > cols[0]:
                be.forward_xslab(self.proj_mat.poses, self.d_rec, self.d_ax, cols[0], cols[1])

        for i, (_, (x_lo, x_hi), cols) in enumerate(self._plan):
            o, n = x_lo * plane, (x_hi - x_lo) * plane
            piece, tail = self._pieces(n)
            if piece:
                comm.wait_next()                                    # this slab's reduce-scatter: my piece of d_bp is final
                upd(o + r * piece, piece, True)                     # 1/P of the slab: V scaling, positivity, error sum
            if tail:
                comm.wait_next()                                    # the tail's all-reduce
                upd(o + piece * P, tail, r == 0 or not any_piece)   # identical on every rank; its error counted once
            if piece:
                comm.allgather_async(self.d_rec.view(o, piece * P), piece)      # after the update (stream order -> communication stream)
                gathers += 1
                # software-pipelined: slab i's pieces travel while slab i + 1 is updated; the forward projection that needs slab i - 1
                # whole waits for ITS all-gather only
                if gathers > 1:
                    comm.wait_next_gather()
                    forward_behind(fwd_after)
                fwd_after = cols
            else:
                forward_behind(cols)                                # all-reduced slab: final on every rank as soon as it is updated
        if gathers:
            comm.wait_next_gather()
            forward_behind(fwd_after)
        comm.join()                                 # nothing left pending (bookkeeping; every collective has been waited for)
        self._ax_ready = ahead
        if self.d_gt is None:
            return None
        err = be.update_acc_fetch()
        # with pieces every rank summed (gt - rec)^2 over its own voxels only (and rank 0 over the tails): one more scalar all-reduce,
        # issued by every rank alike (whether a ground truth is given is a property of the run, not of a rank)
        return self._allreduce_scalar(err) if any_piece else err

    def iterate_device(self, niter=100, positivity=False, projections=None, debug=False):
        self._ax_ready = False                      # a projection made ahead never outlives the call that made it
        return super(SIRT, self).iterate_device(niter=niter, positivity=positivity, projections=projections, debug=debug)

    def _initialize(self):
        self._zero_guard = 1.e-8      # sirt_mpi.py:69-70
        self._stop_after = 1          # sirt_mpi.py:116
        self._pipelined = self._iter_pipelined = False
        self._ax_ready = False
        super(SIRT, self)._initialize()
        self._decide_pipeline(self.d_res, self.d_bp)
        self._ax_ready = False

    def run_main_iteration(self, niter=100, positivity=False, make_plot=False, debug=False):
        return super(SIRT, self).run_main_iteration(niter=niter, make_plot=make_plot, positivity=positivity, debug=debug)
