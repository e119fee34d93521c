"""
Device backend used by the operator and the solvers: a thin object over the C-ABI (include/tomo.h).

The solvers (recon/sirt.py, recon/cgls.py, their sharded twins) are written against this small
interface -- buffers in, buffers out, scalars back -- so that their control flow (stop rules, guards,
re-initialisation) can be exercised in CPU-only tests by injecting a stand-in with the same methods
(tests/backends.py).  The product only ever constructs HipBackend; nothing here falls back to CPU.
"""
import ctypes

import numpy as np

try:
    from . import _lib
except ImportError:      # package directory itself on sys.path (reference-style `import utilities` layout)
    import _lib

_c_vp = ctypes.c_void_p


class HipBackend(object):
    name = "hip"

    def __init__(self, geometry, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.Context()
        self.lib = self.ctx.lib
        self.geometry = geometry
        self.n_vox = int(np.prod(geometry.vox_shape))
        self.n_det = int(np.prod(geometry.det_shape))
        self.ctx.set_geometry(geometry)
        self.comm = None

    # ---- buffers
    def upload(self, host):
        return self.ctx.to_device(np.asarray(host).ravel(), np.float32)

    def download(self, buf):
        return buf.download()

    def zeros(self, n):
        return self.ctx.zeros((int(n),), np.float32)

    def empty(self, n):
        return self.ctx.empty((int(n),), np.float32)

    def copy(self, dst, src):
        dst.copy_from(src)

    def is_buffer(self, x):
        return isinstance(x, _lib.DeviceArray)

    def _geom(self):
        self.ctx.set_geometry(self.geometry)     # no-op unless another operator changed it

    # ---- projectors (poses: (n,7) float64, see _lib.poses_array)
    def forward(self, poses, vol, out):
        self._geom()
        n = poses.shape[0]
        if vol.size != self.n_vox or out.size != n * self.n_det:
            raise ValueError("forward: buffer sizes do not match geometry")
        self.ctx.check(self.lib.tomo_forward(self.ctx.handle, _lib.dptr(poses), n, vol.ptr, out.ptr))
        return out

    def adjoint(self, poses, proj, out, accumulate=False):
        self._geom()
        n = poses.shape[0]
        if out.size != self.n_vox or proj.size != n * self.n_det:
            raise ValueError("adjoint: buffer sizes do not match geometry")
        self.ctx.check(self.lib.tomo_adjoint(self.ctx.handle, _lib.dptr(poses), n, proj.ptr, out.ptr, 1 if accumulate else 0))
        return out

    def xslab_info(self):
        """(number of x tile columns, their width in voxels) of the tile adjoint."""
        self._geom()
        n, w = ctypes.c_int(0), ctypes.c_int(0)
        self.ctx.check(self.lib.tomo_adjoint_xslab_info(self.ctx.handle, ctypes.byref(n), ctypes.byref(w)))
        return n.value, w.value

    def adjoint_xslab(self, poses, proj, out, xt0, xt1, same_sinogram=False):
        """A^T y restricted to the x tile columns [xt0, xt1); ADDS into `out` (zero it first).  Raises TomoError for poses
        the tile kernels decline.  same_sinogram: the caller vouches that `proj` is unchanged since the previous back-projection
        call (the later slabs of one pass) -- its non-empty detector planes are then not looked for again."""
        self._geom()
        if same_sinogram:
            self.ctx.set_option("reuse_sino_flags", 1)
        try:
            self.ctx.check(self.lib.tomo_adjoint_xslab(self.ctx.handle, _lib.dptr(poses), poses.shape[0], proj.ptr, out.ptr, int(xt0), int(xt1)))
        finally:
            if same_sinogram:
                self.ctx.set_option("reuse_sino_flags", 0)

    def forward_xslab(self, poses, vol, out, xt0, xt1):
        """The partial ray sums of the x tile columns [xt0, xt1) (they read only the voxels x in [w*xt0 - 1, w*xt1]); ADDS into
        `out` (zero it first).  Raises TomoError for poses the tile kernels decline."""
        self._geom()
        self.ctx.check(self.lib.tomo_forward_xslab(self.ctx.handle, _lib.dptr(poses), poses.shape[0], vol.ptr, out.ptr, int(xt0), int(xt1)))

    def tiles_take(self, poses, proj, vol):
        """True when every pose of `poses` takes the tile kernels, i.e. the x-slab forms above exist for them (an empty
        column range launches nothing)."""
        try:
            self.adjoint_xslab(poses, proj, vol, 0, 0)
            self.forward_xslab(poses, vol, proj, 0, 0)      # the pipelined update also projects slab by slab: needs fwd_variant 3 as well
            return True
        except _lib.TomoError as e:
            if "do not take the tile kernels" not in str(e):
                raise
            return False

    def backproject_voxel(self, poses, det, out):
        self._geom()
        n = poses.shape[0]
        if out.size != self.n_vox or det.size != n * self.n_det:
            raise ValueError("backproject_voxel: buffer sizes do not match geometry")
        self.ctx.check(self.lib.tomo_backproject_voxel(self.ctx.handle, _lib.dptr(poses), n, det.ptr, out.ptr))
        return out

    def proj_grad(self, pose, vol, proj_out, grad_out, row_order=0):
        self._geom()
        self.ctx.check(self.lib.tomo_proj_grad(self.ctx.handle, _lib.dptr(pose), vol.ptr, proj_out.ptr, grad_out.ptr, row_order))

    def cost_grad(self, poses, vol, b, resid=None, rows=None):
        """-> (cost[n], grad6[n,6]) float64 on the host (fused residual / cost / gradient reduction).  With `rows`, pose i
        is compared with row rows[i] of the device-resident table `b` ([n_rows][n_det]) instead of row i."""
        self._geom()
        n = poses.shape[0]
        cost = np.zeros(n, np.float64)
        g6 = np.zeros((n, 6), np.float64)
        if rows is None:
            if b.size < n * self.n_det:
                raise ValueError("cost_grad: measured projections smaller than n * n_det")
            self.ctx.check(self.lib.tomo_cost_grad(self.ctx.handle, _lib.dptr(poses), n, vol.ptr, b.ptr, _lib.dptr(cost), _lib.dptr(g6),
                                                   resid.ptr if resid is not None else None))
        else:
            rows = np.ascontiguousarray(rows, np.int32)
            if rows.size != n or b.size % self.n_det:
                raise ValueError("cost_grad: rows must have one entry per pose and b must be whole rows")
            self.ctx.check(self.lib.tomo_cost_grad_rows(self.ctx.handle, _lib.dptr(poses), n, vol.ptr, b.ptr,
                                                        rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), b.size // self.n_det,
                                                        _lib.dptr(cost), _lib.dptr(g6), resid.ptr if resid is not None else None))
        return cost, g6

    def csr_assemble(self, poses, mask=None, precision=np.float32, max_nnz=None):
        """The reference's assembled projection matrix (utilities/projection_operators.py:54-76) built on the device: triplets of all
        projections, mask filter, sort, duplicate sums, row pointers -> (data [precision], indices int32, indptr int64).
        max_nnz: refuse (MemoryError) BEFORE any host array is allocated or a byte is downloaded; the device copy is dropped on refusal
        and on any failure between assembly and download."""
        self._geom()
        bits = 64 if np.dtype(precision) == np.float64 else 32
        nnz = ctypes.c_int64(0)
        poses = np.ascontiguousarray(poses, np.float64)
        self.ctx.check(self.lib.tomo_csr_assemble(self.ctx.handle, _lib.dptr(poses), poses.shape[0], mask.ptr if mask is not None else None, bits,
                                                  ctypes.byref(nnz)))
        try:
            if max_nnz is not None and nnz.value > max_nnz:
                raise MemoryError("tocsr: %d entries, more than %d; keep the operator matrix-free" % (nnz.value, max_nnz))
            data = np.empty(nnz.value, np.float64 if bits == 64 else np.float32)
            indices = np.empty(nnz.value, np.int32)
            indptr = np.empty(poses.shape[0] * self.n_det + 1, np.int64)
        except BaseException:
            self.lib.tomo_csr_fetch(self.ctx.handle, None, None, None)      # drop the device copy (tens of GB at 128^3)
            raise
        self.ctx.check(self.lib.tomo_csr_fetch(self.ctx.handle, data.ctypes.data_as(_c_vp), indices.ctypes.data_as(_c_vp), indptr.ctypes.data_as(_c_vp)))
        return data, indices, indptr

    def triplets(self, pose):
        """COO triplets (dat_inds, det_inds, wts float64) of one projection in the emission order of
        src/ray_wt_grad.f90:1-92 (small volumes only)."""
        self._geom()
        n = ctypes.c_int64(0)
        self.ctx.check(self.lib.tomo_triplets(self.ctx.handle, _lib.dptr(pose), 0, None, None, None, ctypes.byref(n)))
        dat, det, wts = np.empty(n.value, np.int32), np.empty(n.value, np.int32), np.empty(n.value, np.float64)
        if n.value:
            self.ctx.check(self.lib.tomo_triplets(self.ctx.handle, _lib.dptr(pose), n.value, dat.ctypes.data_as(_c_vp),
                                                  det.ctypes.data_as(_c_vp), wts.ctypes.data_as(_c_vp), ctypes.byref(n)))
        return dat, det, wts

    def vox_splat(self, pose, cor3, vol, img_out, grad_out=None):
        self._geom()
        cor3 = np.ascontiguousarray(cor3, np.float64).reshape(3)
        self.ctx.check(self.lib.tomo_vox_splat(self.ctx.handle, _lib.dptr(pose), _lib.dptr(cor3), vol.ptr, img_out.ptr,
                                               grad_out.ptr if grad_out is not None else None))

    def vox_triplets(self, pose, cor3):
        """(dat_inds, det_inds, wts float32) of src/vox_wt_grad.f90:58-112 bilinear_sparse, in its emission order."""
        self._geom()
        cor3 = np.ascontiguousarray(cor3, np.float64).reshape(3)
        det4 = np.empty(4 * self.n_vox, np.int32)
        w4 = np.empty(4 * self.n_vox, np.float32)
        self.ctx.check(self.lib.tomo_vox_triplets(self.ctx.handle, _lib.dptr(pose), _lib.dptr(cor3), det4.ctypes.data_as(_c_vp),
                                                  w4.ctypes.data_as(_c_vp)))
        keep = det4 >= 0
        dat = (np.arange(4 * self.n_vox, dtype=np.int64) // 4).astype(np.int32)
        return dat[keep], det4[keep], w4[keep]

    def phantom(self, out, shape, table):
        """Fill `out` with the ellipsoid phantom of utilities/generate_phantom (generated on the device)."""
        table = np.ascontiguousarray(table, np.float64)
        nx, ny, nz = (int(v) for v in shape)
        self.ctx.check(self.lib.tomo_phantom_ellipsoids(self.ctx.handle, out.ptr, nx, ny, nz, _lib.dptr(table), table.shape[0]))
        return out

    # ---- solver vector kernels
    def fill(self, buf, value):
        self.ctx.check(self.lib.tomo_vec_fill(self.ctx.handle, buf.ptr, buf.size, float(value)))

    def recip_guard(self, buf, thresh=None):
        """x -> 1/x with x==0 -> 0 (thresh None; recon/sirt.py:37-40) or x<thresh -> 0 (recon/sirt_mpi.py:69-72)."""
        strict = thresh is None
        self.ctx.check(self.lib.tomo_vec_recip_guard(self.ctx.handle, buf.ptr, buf.size, 0.0 if strict else float(thresh), 1 if strict else 0))

    def residual_scale(self, b, ax, w, out):
        s = ctypes.c_double(0)
        self.ctx.check(self.lib.tomo_vec_residual_scale(self.ctx.handle, b.ptr, ax.ptr, w.ptr if w is not None else None, out.ptr,
                                                        b.size, ctypes.byref(s)))
        return s.value

    def update(self, rec, bp, v, positivity=False, gt=None):
        s = ctypes.c_double(0)
        self.ctx.check(self.lib.tomo_vec_update(self.ctx.handle, rec.ptr, bp.ptr, v.ptr if v is not None else None, rec.size,
                                                1 if positivity else 0, gt.ptr if gt is not None else None, ctypes.byref(s)))
        return s.value if gt is not None else None

    def update_acc(self, rec, bp, v, positivity=False, gt=None, first=True):
        """`update` on one x slab of a pipelined iteration: nothing comes back to the host; ||gt - rec||^2 accumulates on the
        device across the slabs (restarted when `first`) until update_acc_fetch()."""
        self.ctx.check(self.lib.tomo_vec_update_acc(self.ctx.handle, rec.ptr, bp.ptr, v.ptr if v is not None else None, rec.size,
                                                    1 if positivity else 0, gt.ptr if gt is not None else None, 1 if first else 0))

    def update_acc_fetch(self):
        s = ctypes.c_double(0)
        self.ctx.check(self.lib.tomo_vec_update_acc_fetch(self.ctx.handle, ctypes.byref(s)))
        return s.value

    def axpy(self, y, x, a):
        self.ctx.check(self.lib.tomo_vec_axpy(self.ctx.handle, y.ptr, x.ptr, float(a), y.size))

    def xpay(self, y, x, a):
        self.ctx.check(self.lib.tomo_vec_xpay(self.ctx.handle, y.ptr, x.ptr, float(a), y.size))

    def sub(self, out, a, b):
        self.ctx.check(self.lib.tomo_vec_sub(self.ctx.handle, out.ptr, a.ptr, b.ptr, out.size))

    def mul(self, y, x):
        self.ctx.check(self.lib.tomo_vec_mul(self.ctx.handle, y.ptr, x.ptr, y.size))

    def dot(self, a, b):
        s = ctypes.c_double(0)
        self.ctx.check(self.lib.tomo_vec_dot(self.ctx.handle, a.ptr, b.ptr, a.size, ctypes.byref(s)))
        return s.value

    def diff_sumsq(self, a, b):
        s = ctypes.c_double(0)
        self.ctx.check(self.lib.tomo_vec_diff_sumsq(self.ctx.handle, a.ptr, b.ptr, a.size, ctypes.byref(s)))
        return s.value

    # ---- device accumulators: several scalars of an iteration, one host synchronisation (include/tomo.h tomo_acc_*)
    N_ACC = 16

    def acc_zero(self, slot0, n=1):
        self.ctx.check(self.lib.tomo_acc_zero(self.ctx.handle, int(slot0), int(n)))

    def dot_acc(self, a, b, slot, diff=False):
        """slot += sum(a*b)  (diff: sum((a-b)^2)); nothing comes back to the host."""
        if a.size != b.size:
            raise ValueError("dot_acc: operand sizes differ")
        self.ctx.check(self.lib.tomo_vec_dot_acc(self.ctx.handle, a.ptr, b.ptr, a.size, 1 if diff else 0, int(slot)))

    def acc_fetch(self, slot0, n=1, allreduce=False):
        """-> the n accumulators from slot0 on; allreduce: summed over the ranks of the context's RCCL communicator first (one small
        device-side collective)."""
        out = np.zeros(int(n), np.float64)
        self.ctx.check(self.lib.tomo_acc_fetch(self.ctx.handle, int(slot0), int(n), 1 if allreduce else 0, _lib.dptr(out)))
        return out

    # ---- regularised solvers' vector kernels (recon/regularized.py:433, utilities/tv_denoise.py:98-170)
    def soft_threshold(self, out, x, lam):
        self.ctx.check(self.lib.tomo_vec_soft_threshold(self.ctx.handle, out.ptr, x.ptr, x.size, float(lam)))
        return out

    def tv_denoise_fista(self, im, out, shape, weight=50, niter=200, eps=1.e-5, check_gap_frequency=3):
        """-> (iterations done, last dual gap); `out` holds the reference's return value."""
        nx, ny, nz = (int(v) for v in shape)
        it, gap = ctypes.c_int(0), ctypes.c_double(0)
        self.ctx.check(self.lib.tomo_tv_denoise_fista(self.ctx.handle, im.ptr, out.ptr, nx, ny, nz, float(weight), int(niter), float(eps),
                                                      int(check_gap_frequency), ctypes.byref(it), ctypes.byref(gap)))
        return it.value, gap.value

    def tv_norm_3d(self, x, shape):
        nx, ny, nz = (int(v) for v in shape)
        v = ctypes.c_double(0)
        self.ctx.check(self.lib.tomo_tv_norm_3d(self.ctx.handle, x.ptr, nx, ny, nz, ctypes.byref(v)))
        return v.value

    def sync(self):
        self.ctx.sync()
