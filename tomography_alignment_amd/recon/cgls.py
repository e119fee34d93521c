"""
CGLS on the normal equations with the constructor / run_main_iteration signature of the reference's
recon/cgls.py:7-104, device-resident.  The reference file does not import in its own snapshot
(`utilities.linear_operators` is missing, :3; `object['precision']`, :20; `self.method` undefined,
:51); what is restated here is its sparse-matrix branch (:54-82), which is the only branch whose
operands exist.  Re-initialisation rule: if ||b - A rec|| rises, undo the step and restart from the
current `rec`; two consecutive restarts quit (:60-68).
"""
import numpy as np

try:
    from ..utilities import projection_operators
except ImportError:      # imported as top-level `recon`
    from utilities import projection_operators


class CGLS(object):

    def __init__(self, geometry, projections, angles, xyz_shift, options={}):
        self.geometry = geometry
        self.projections = projections
        self.angles = angles
        self.xyz_shift = xyz_shift
        self.n_proj = angles.shape[0]
        self.ground_truth = options['ground_truth'] if 'ground_truth' in options else None
        self.rec = options['rec'] if 'rec' in options else None
        if self.rec is None:
            self.rec = np.zeros((int(self.geometry.n_vox),), dtype=getattr(self.projections, 'dtype', np.float32))
        self.precision = options['precision'] if 'precision' in options else np.float32
        self._backend = options.get('_backend')
        self.rms_error = None
        self.f_proj_obj = None
        self.proj_mat = None
        self._bufs = None
        self._initialize()

    def _my_rows(self):
        return np.arange(self.n_proj)

    def _allreduce_vol(self, buf):
        return buf

    def _allreduce_scalar(self, v):
        return v

    def _local_geometry(self, rows):
        return self.geometry

    def _initialize(self):
        rows = self._rows = self._my_rows()
        if self.f_proj_obj is None:
            self.f_proj_obj = projection_operators.ProjectionMatrix(self._local_geometry(rows), precision=self.precision,
                                                                    backend=self._backend)
            self.proj_mat = self.f_proj_obj.projection_matrix(phi=self.angles[rows, 0], alpha=self.angles[rows, 1],
                                                              beta=self.angles[rows, 2], xyz_shift=self.xyz_shift[rows])
        be = self.be = self.f_proj_obj.backend
        n_vox, n_rows = be.n_vox, rows.size * be.n_det
        if self._bufs is None:
            self._bufs = True
            if be.is_buffer(self.projections):          # already in HBM (this rank's rows): no PCIe traffic (bench.py at 1024^3)
                self.d_b = self.projections
            else:
                self.d_b = be.upload(np.asarray(self.projections, np.float32).reshape(self.n_proj, -1)[rows])
            self.d_tmp = be.empty(n_rows)               # A rec of the ||b - A rec|| test (recon/cgls.py:58-59), kept across iterations
            self.d_rec = be.upload(np.asarray(self.rec, np.float32).ravel())
            self.d_r = be.empty(n_rows)
            self.d_q = be.empty(n_rows)
            self.d_p = be.empty(n_vox)
            self.d_s = be.empty(n_vox)
            self.d_gt = None if self.ground_truth is None else be.upload(np.asarray(self.ground_truth, np.float32).ravel())
        self.proj_mat.apply(self.d_rec, self.d_q)
        be.sub(self.d_r, self.d_b, self.d_q)                                 # _r = b - A rec        cgls.py:33
        self._allreduce_vol(self.proj_mat.T.apply(self.d_r, self.d_p))       # _p = A^T _r           :34 ; cgls_mpi.py:55
        self._gamma = be.dot(self.d_p, self.d_p)                             # ||_p||^2              :36

    def iterate_device(self, niter=100):
        """run_main_iteration without the final download: `self.d_rec` holds the reconstruction; returns (k, rms_error[:k])."""
        rec, rms = self.run_main_iteration(niter=niter, _download=False)
        return len(rms), rms

    def run_main_iteration(self, make_plot=False, niter=100, debug=False, _download=True):
        be = self.be
        if self.ground_truth is None:
            if be.is_buffer(self.projections):
                norm_factor = np.sqrt(self._allreduce_scalar(be.dot(self.d_b, self.d_b)))
            else:
                norm_factor = np.linalg.norm(np.asarray(self.projections, np.float32))
        else:
            norm_factor = np.linalg.norm(self.ground_truth)
        k, reinit_iter = 0, 0
        conv = np.zeros((niter,))
        self.rms_error = np.zeros((niter,))
        while k < niter:
            self._apply_to_p()                                                               # r = A p       :54
            qq, conv_sumsq = self._q_norms()                                                 # :56 (cgls_mpi.py:72-76: the monitor too)
            alpha = self._gamma / qq
            be.axpy(self.d_rec, self.d_p, alpha)                                             # :57
            if conv_sumsq is None:
                conv_sumsq = self._conv_sumsq()                                              # :58-59: ||b - A rec|| of the UPDATED rec
            conv[k] = np.sqrt(conv_sumsq)
            if k > 0 and conv[k] > conv[k - 1]:
                print('reinitializing at iteration %d' % k)
                if reinit_iter + 1 == k:
                    print('need to re-initialize at two consecutive iterations: quitting')
                    self.rec = be.download(self.d_rec) if _download else self.rec
                    return self.rec, self.rms_error[:k]
                be.axpy(self.d_rec, self.d_p, -alpha)                                        # :66
                q_keep = be.empty(self.d_q.size)
                be.copy(q_keep, self.d_q)
                self._initialize()                                                           # :67
                be.copy(self.d_q, q_keep)
                reinit_iter = k
            be.axpy(self.d_r, self.d_q, -alpha)                                              # _r -= alpha r  :70
            gamma, rr = self._backproject_and_norms(self.ground_truth is None)               # p = A^T _r :72 ; ||p||^2 ; ||_r||^2 (:80)
            beta = gamma / self._gamma
            self._gamma = gamma
            self._update_p(beta, last=(k + 1 >= niter))                                      # _p = p + beta _p   :78
            if self.ground_truth is None:
                self.rms_error[k] = np.sqrt(rr) / norm_factor                                # :80
            else:
                self.rms_error[k] = np.sqrt(be.diff_sumsq(self.d_rec, self.d_gt)) / norm_factor                # :82
            k += 1
        self.rec = be.download(self.d_rec) if _download else self.rec
        return self.rec, self.rms_error[:k]

    # ---- the steps the angle-sharded subclass replaces (recon/cgls_mpi.py)
    def _apply_to_p(self):
        """d_q = A d_p."""
        self.proj_mat.apply(self.d_p, self.d_q)

    def _q_norms(self):
        """(||A p||^2, the convergence monitor's sum of squares when it is a function of A p alone -- recon/cgls_mpi.py:74 -- else None:
        recon/cgls.py:58 monitors ||b - A rec|| of the rec updated with this very alpha, see _conv_sumsq)."""
        return self.be.dot(self.d_q, self.d_q), None

    def _backproject_and_norms(self, want_rr):
        """d_s = A^T d_r; -> (||d_s||^2, ||d_r||^2 or None)."""
        be = self.be
        self.proj_mat.T.apply(self.d_r, self.d_s)
        return be.dot(self.d_s, self.d_s), (be.dot(self.d_r, self.d_r) if want_rr else None)

    def _update_p(self, beta, last=False):
        self.be.xpay(self.d_p, self.d_s, beta)

    def _conv_sumsq(self):
        """||b - A rec||^2 (recon/cgls.py:58-59); costs one extra forward projection, as in the reference."""
        be = self.be
        self.proj_mat.apply(self.d_rec, self.d_tmp)
        return be.diff_sumsq(self.d_b, self.d_tmp)
