"""
SIRT with the constructor / run_main_iteration signature of the reference's recon/sirt.py:7-107,
running device-resident: `rec`, `b`, `W`, `V`, the residual and the back-projection stay in HBM across
iterations; per iteration only two scalars cross PCIe.

    W = 1/(A.1), V = 1/(A^T.1)  (0 -> 0)                 recon/sirt.py:33-40
    rec += V * A^T( W * (b - A rec) ) ; positivity         :59-67
    rms_error[k] = ||gt - rec||/||gt||  or  ||res||/||b||  :69-73
    stop when rms_error rises (k > 0)                      :75-78

(The reference's own recon/sirt.py also runs unmodified on this package's operator, through the scipy
unbound-call protocol of utilities/projection_operators.RayOperator -- tests/test_gpu_solvers.py.)
"""
import time

import numpy as np

try:
    from ..utilities import projection_operators
except ImportError:      # imported as top-level `recon` (package directory on sys.path, like the reference tree)
    from utilities import projection_operators


class SIRT(object):

    def __init__(self, geometry, projections, angles, xyz_shifts, options={}):
        self.geometry = geometry
        self.projections = projections
        self.angles = angles
        self.xyz_shifts = xyz_shifts
        self.n_proj = angles.shape[0]
        self.ground_truth = options['ground_truth'] if 'ground_truth' in options else None
        self.rec = options['rec'] if 'rec' in options else None
        self._rec_given = self.rec is not None
        if self.rec is None:
            self.rec = np.zeros((int(self.geometry.n_vox),), dtype=getattr(self.projections, 'dtype', np.float32))
        self.precision = options['precision'] if 'precision' in options else np.float32
        self.voxel_mask = options['voxel_mask'] if 'voxel_mask' in options else None
        self._backend = options.get('_backend')          # test seam; None -> HipBackend (no CPU fallback)
        self.f_proj_obj = None
        self.proj_mat = None
        # reference quirks that differ between recon/sirt.py and recon/sirt_mpi.py; the sharded subclass flips them
        self._zero_guard = None      # None: == 0 (sirt.py:37-38) ; float: < thresh (sirt_mpi.py:69-70)
        self._stop_after = 0         # stop test needs k > 0 (sirt.py:75) / k > 1 (sirt_mpi.py:116)
        self._initialize()

    # ---- hooks the sharded subclass overrides
    def _my_rows(self):
        return np.arange(self.n_proj)

    def _allreduce_vol(self, buf):
        return buf

    def _allreduce_scalar(self, v):
        return v

    def _allreduce_host(self, a):
        """Sum of a HOST array over the ranks (identity here)."""
        return a

    def _is_root(self):
        return True

    def _initialize(self):
        rows = self._my_rows()
        self._rows = rows
        if self.f_proj_obj is None:
            self.f_proj_obj = projection_operators.ProjectionMatrix(self._local_geometry(rows), precision=self.precision,
                                                                    backend=self._backend)
            self.proj_mat = self.f_proj_obj.projection_matrix(phi=self.angles[rows, 0], alpha=self.angles[rows, 1],
                                                              beta=self.angles[rows, 2], xyz_shift=self.xyz_shifts[rows],
                                                              voxel_mask=self.voxel_mask)
        be = self.be = self.f_proj_obj.backend
        n_vox, n_rows = be.n_vox, rows.size * be.n_det
        ones_v = be.empty(n_vox)
        be.fill(ones_v, 1.0)
        self.d_W = self.proj_mat.apply(ones_v, be.empty(n_rows))                 # sirt.py:33
        ones_p = be.empty(n_rows)
        be.fill(ones_p, 1.0)
        self.d_V = self._allreduce_vol(self.proj_mat.T.apply(ones_p, ones_v))    # sirt.py:34 ; sirt_mpi.py:68
        be.recip_guard(self.d_V, self._zero_guard)                               # sirt.py:37-40
        be.recip_guard(self.d_W, self._zero_guard)
        self.d_res = ones_p                                                      # scratch [n_rows]
        self.d_ax = be.empty(n_rows)
        self.d_bp = be.empty(n_vox)
        if not self._rec_given:
            self.d_rec = be.zeros(n_vox)              # zero start (sirt.py:18-19) without a host round trip
        else:
            self.d_rec = self.rec if be.is_buffer(self.rec) else be.upload(np.asarray(self.rec, np.float32).ravel())
        self.d_b = None
        self.d_gt = None

    def _local_geometry(self, rows):
        return self.geometry

    @property
    def W(self):
        return self.be.download(self.d_W)

    @property
    def V(self):
        return self.be.download(self.d_V)

    def _prepare(self, projections=None):
        be = self.be
        if projections is not None:
            self.projections = projections
            self.d_b = None
        if be.is_buffer(self.projections):              # already in HBM (this rank's rows): no PCIe traffic
            if self.d_b is None:
                self.d_b = self.projections
            b_sumsq = self._allreduce_scalar(be.dot(self.d_b, self.d_b))
        else:
            b_all = np.asarray(self.projections, np.float32).reshape(self.n_proj, -1)
            if self.d_b is None:
                self.d_b = be.upload(b_all[self._rows])
            b_sumsq = float(np.linalg.norm(b_all)) ** 2
        if self.ground_truth is not None:
            if be.is_buffer(self.ground_truth):
                self.d_gt = self.ground_truth
                return np.sqrt(be.dot(self.d_gt, self.d_gt))
            self.ground_truth = np.asarray(self.ground_truth).ravel()
            if self.d_gt is None:
                self.d_gt = be.upload(self.ground_truth.astype(np.float32))
            return np.linalg.norm(self.ground_truth)                                   # sirt.py:47-49
        return np.sqrt(b_sumsq)                                                        # sirt.py:51

    def iterate_device(self, niter=100, positivity=False, projections=None, debug=False):
        """The loop of recon/sirt.py:58-105 with every vector in HBM; returns (k, rms_error[:k]).
        `self.d_rec` holds the reconstruction afterwards."""
        be = self.be
        norm_factor = self._prepare(projections)
        stop, k = 0, 0
        rms_error = np.zeros((niter,))
        convergence = np.zeros((niter,))
        t_start = time.time()
        while k < niter and not stop:
            self._forward()                                                             # sirt.py:59
            sumsq = be.residual_scale(self.d_b, self.d_ax, self.d_W, self.d_res)        # :60-61 (W * res) and :69
            self._backproject_scaled()                                                  # :61,63 ; sirt_mpi.py:98-103
            err = self._update(positivity, last=(k + 1 >= niter))                       # :64-67,73
            convergence[k] = np.sqrt(self._allreduce_scalar(sumsq))                     # :69 ; sirt_mpi.py:110
            rms_error[k] = convergence[k] / norm_factor if self.d_gt is None else np.sqrt(err) / norm_factor
            if k > self._stop_after and rms_error[k] > rms_error[k - 1]:
                stop = 1
                if self._is_root():
                    print('semi-convergence criterion reached: stopping at k %3d with RMSE = %4.5f' % (k, rms_error[k]))
            if k > 0 and k % 20 == 0 and debug:
                print('time taken for 20 SIRT iterations = %4.5f' % (time.time() - t_start))
                t_start = time.time()
            k += 1
        self.rms_error = rms_error
        self.convergence = convergence
        return k, rms_error[:k]

    def _forward(self):
        """d_ax = A d_rec."""
        self.proj_mat.apply(self.d_rec, self.d_ax)

    def _update(self, positivity, last=False):
        """rec += V * d_bp, clamp; returns ||gt - rec||^2 when a ground truth is given."""
        return self.be.update(self.d_rec, self.d_bp, self.d_V, positivity, self.d_gt)

    def _backproject_scaled(self):
        """d_bp = A^T d_res, summed over the angle shards.  The reference scales by V here (sirt.py:63) and, sharded, BEFORE its Allreduce
        (sirt_mpi.py:101-103); V is the same on every rank, so the scaling commutes with the sum and is applied by the update
        (`rec += V * bp` in one pass) -- one read-modify-write of the volume per iteration less."""
        self.proj_mat.T.apply(self.d_res, self.d_bp)                                    # sirt.py:61
        self._allreduce_vol(self.d_bp)                                                  # sirt_mpi.py:102-103

    def run_regularized_gradient_descent(self, niter=100, reg_param=1.0, positivity=True, make_plot=False, debug=False):
        """Tikhonov-regularised least squares by gradient descent, step by scipy's strong-Wolfe `optimize.line_search` on
        `my_f` / `my_fp` below -- the reference's recon/sirt.py:109-180, same defaults, same stop rule (k > 1 and rms rising),
        same fallback step 1e-3 when the line search gives up.  A secondary method: the iterate lives on the HOST (scipy's line
        search does its own arithmetic on numpy arrays), every function / gradient evaluation is one forward (+ one
        back-projection) through the operator with the volume crossing PCIe -- fine at the sizes it is used at; the
        device-resident solver is run_main_iteration / iterate_device."""
        from scipy import optimize
        if make_plot:
            print('make_plot is not supported on the device-resident solver; ignoring')
        A = self.proj_mat
        n_rows = np.size(self._rows)
        if self.be.is_buffer(self.projections):
            b = np.asarray(self.be.download(self.projections), np.float32).reshape(n_rows, -1)      # this rank's rows, already in HBM
        else:
            b = np.asarray(self.projections, np.float32).reshape(self.n_proj, -1)[self._rows]
        # angle-sharded subclass: A and b are this rank's rows; the data terms are summed over the ranks (recon/sirt_mpi.py:160-178)
        host_sum, scalar_sum = self._allreduce_host, self._allreduce_scalar

        def f(x, A_, b_, lam):
            r_ = A_.dot(np.ravel(x)) - np.ravel(b_)
            return scalar_sum(0.5 * float(np.linalg.norm(r_)) ** 2) + 0.5 * lam * np.linalg.norm(x) ** 2

        def fp(x, A_, b_, lam):
            r_ = A_.dot(np.ravel(x)) - np.ravel(b_)
            return host_sum(A_.T.dot(r_)) + lam * x
        sharded = type(self)._allreduce_host is not SIRT._allreduce_host       # unsharded: the module-level my_f / my_fp, as the reference
        rec = np.array(self.be.download(self.d_rec), np.float32)                      # sirt.py:19 (the current reconstruction)
        if self.ground_truth is not None:
            gt = np.asarray(self.ground_truth if not self.be.is_buffer(self.ground_truth) else self.be.download(self.ground_truth)).ravel()
            norm_factor = np.linalg.norm(gt)                                           # :115-117
        else:
            gt, norm_factor = None, np.sqrt(self._allreduce_scalar(float(np.linalg.norm(b)) ** 2))      # :119
        stop, k = 0, 0
        rms_error, convergence = np.zeros((niter,)), np.zeros((niter,))
        while k < niter and not stop:
            res = b - A.dot(rec).reshape(n_rows, -1)                                   # :128-129
            grad = -host_sum(A.T.dot(res.ravel())) + reg_param * rec                   # :130,132   A^T(A x - b) + lambda x
            alpha = optimize.line_search(f if sharded else my_f, fp if sharded else my_fp, rec, -grad, args=(A, b, reg_param))[0]      # :135-137
            if alpha is None:
                alpha = 1.e-3                                                          # :138-139
            rec -= alpha * grad                                                        # :142
            if positivity:
                rec[rec < 0.] = 0.                                                     # :145-146
            convergence[k] = np.sqrt(self._allreduce_scalar(float(np.linalg.norm(res)) ** 2))
            rms_error[k] = convergence[k] / norm_factor if gt is None else np.linalg.norm(gt - rec) / norm_factor      # :148-152
            if k > 1 and rms_error[k] > rms_error[k - 1]:                              # :154
                stop = 1
                if self._is_root():
                    print('semi-convergence criterion reached: stopping at k %3d with RMSE = %4.5f' % (k, rms_error[k]))
            k += 1
        self.rec = rec
        self.d_rec.upload(rec)
        self.rms_error, self.convergence = rms_error, convergence
        return rec.reshape(tuple(int(v) for v in self.geometry.vox_shape)), rms_error[:k]

    def run_main_iteration(self, niter=100, make_plot=False, projections=None, positivity=False, debug=False):
        if make_plot:
            print('make_plot is not supported on the device-resident solver; ignoring')
        k, rms = self.iterate_device(niter=niter, positivity=positivity, projections=projections, debug=debug)
        self.rec = self.be.download(self.d_rec)
        return self.rec.reshape(tuple(int(v) for v in self.geometry.vox_shape)), rms


def my_f(x, A, b, _lambda):
    """0.5 |A x - b|^2 + 0.5 lambda |x|^2      (recon/sirt.py:183-189); A: any operator with .dot (a RayOperator, a scipy matrix)."""
    res = A.dot(np.ravel(x)) - np.ravel(b)
    return 0.5 * np.linalg.norm(res) ** 2 + 0.5 * _lambda * np.linalg.norm(x) ** 2


def my_fp(x, A, b, _lambda):
    """A^T (A x - b) + lambda x      (recon/sirt.py:192-197)."""
    res = A.dot(np.ravel(x)) - np.ravel(b)
    return A.T.dot(res) + _lambda * x
