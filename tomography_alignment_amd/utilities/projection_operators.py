"""
Drop-in for the reference's utilities/projection_operators.py:11-122 `ProjectionMatrix`.

`projection_matrix(...)` keeps the reference signature but returns a MATRIX-FREE operator living on
the MI355X instead of a scipy CSR (the reference's CSR is 4.5 GB at 128^3 x 64 angles and cannot
exist at 256^3).  The returned object answers the *unbound* scipy calls the reference's solvers make:

    sparse.csr_matrix.dot(A, x)                                   recon/sirt.py:59
    sparse.csc_matrix.dot(sparse.csr_matrix.transpose(A), y)      recon/sirt.py:61

(scipy's `_spbase.dot` does `self @ other`; `_csr_base.transpose` reads `ndim`, `shape`, `data`,
`indices`, `indptr` and calls `self._csc_container(...)`), so recon/sirt.py runs unmodified.
`projection_gradient(...)` returns `(proj[n_det], grad[6, n_det])` exactly like
utilities/projection_operators.py:112-122, rows tx, ty, tz, phi, alpha, beta.
"""
import numpy as np

try:
    from .. import _lib
    from ..backend import HipBackend
except ImportError:      # imported as top-level `utilities` (package directory on sys.path, like the reference tree)
    import _lib
    from backend import HipBackend


def _normalise_poses(geometry, alpha, beta, phi, xyz_shift):
    """Defaults and n_proj==1 re-wrapping of utilities/projection_operators.py:24-52."""
    if phi is None:
        n_proj = int(geometry.n_proj)
        phi = np.linspace(0., np.pi, n_proj)
    else:
        n_proj = int(np.size(phi))
    alpha = np.zeros(n_proj) if alpha is None else alpha
    beta = np.zeros(n_proj) if beta is None else beta
    xyz_shift = np.zeros((n_proj, 3)) if xyz_shift is None else xyz_shift
    phi = np.asarray(phi, np.float64).reshape(n_proj)
    alpha = np.asarray(alpha, np.float64).reshape(n_proj)
    beta = np.asarray(beta, np.float64).reshape(n_proj)
    xyz_shift = np.asarray(xyz_shift, np.float64).reshape(n_proj, 3)
    return n_proj, phi, alpha, beta, xyz_shift


class RayOperator(object):
    """A (n_proj*n_det) x n_vox linear operator: forward projection A and, as `.T`, its exact adjoint."""

    ndim = 2
    format = "rayop"
    # scipy's csr transpose passes these through to _csc_container; they are never dereferenced
    data = None
    indices = None
    indptr = None

    def __init__(self, backend, poses, precision=np.float32, voxel_mask=None, _adjoint_of=None):
        self.backend = backend
        self.poses = poses
        self.precision = precision
        self.dtype = np.dtype(precision)
        self._is_adjoint = _adjoint_of is not None
        self._fwd = _adjoint_of if self._is_adjoint else self
        n_rows = poses.shape[0] * backend.n_det
        self.shape = (backend.n_vox, n_rows) if self._is_adjoint else (n_rows, backend.n_vox)
        if not self._is_adjoint:
            self._mask = None
            if voxel_mask is not None:
                m = np.asarray(voxel_mask).ravel().astype(bool)
                if m.size != backend.n_vox:
                    raise ValueError("voxel_mask must have n_vox elements")
                if not m.any():
                    print('entire object is masked')      # utilities/projection_operators.py:64-66
                self._mask = backend.upload(m.astype(np.float32))
            self._T = None
            self._scratch = {}

    # -- scipy unbound-method protocol -----------------------------------------------------------
    def _csc_container(self, *args, **kwargs):
        return self.transpose()

    _csr_container = _csc_container

    def transpose(self, axes=None, copy=False):
        if self._is_adjoint:
            return self._fwd
        if self._T is None:
            self._T = RayOperator(self.backend, self.poses, self.precision, _adjoint_of=self)
        return self._T

    @property
    def T(self):
        return self.transpose()

    def __matmul__(self, other):
        return self.dot(other)

    def __mul__(self, other):
        return self.dot(other)

    def matvec(self, x):
        return self.dot(x)

    def rmatvec(self, y):
        return self.transpose().dot(y)

    # -- application -----------------------------------------------------------------------------
    def _buf(self, key, n):
        f = self._fwd
        b = f._scratch.get(key)
        if b is None or b.size != n:
            b = f.backend.empty(n)
            f._scratch[key] = b
        return b

    def apply(self, x_dev, out_dev=None):
        """Device-resident application: DeviceArray in, DeviceArray out (no PCIe traffic)."""
        be, f = self.backend, self._fwd
        if not self._is_adjoint:
            if out_dev is None:
                out_dev = be.empty(self.shape[0])
            src = x_dev
            if f._mask is not None:
                src = self._buf("masked", be.n_vox)
                be.copy(src, x_dev)
                be.mul(src, f._mask)
            return be.forward(self.poses, src, out_dev)
        if out_dev is None:
            out_dev = be.empty(self.shape[0])
        be.adjoint(self.poses, x_dev, out_dev)
        if f._mask is not None:
            be.mul(out_dev, f._mask)
        return out_dev

    def dot(self, other):
        be = self.backend
        if be.is_buffer(other):
            return self.apply(other)
        x = np.asarray(other)
        if x.ndim == 2 and x.shape[1] == 1:
            return self.dot(x[:, 0])[:, np.newaxis]
        if x.size != self.shape[1]:
            raise ValueError("dimension mismatch: operator %s, operand %s" % (self.shape, x.shape))
        xin = self._buf("in%d" % self._is_adjoint, self.shape[1])
        xin.upload(x.ravel())
        out = self.apply(xin, self._buf("out%d" % self._is_adjoint, self.shape[0]))
        res = out.download()
        return res if self.dtype == np.float32 else res.astype(self.dtype)

    def tocsr(self, max_nnz=2 ** 31 - 1, device=True):
        """Materialise the scipy CSR the reference's projection_matrix returns (utilities/projection_operators.py:56-76):
        per-projection COO triplets (emitted on the device with the order and float64 weights of
        src/ray_wt_grad.f90:1-92), weights cast to `precision`, optional voxel-mask filter, duplicates summed,
        explicit zeros kept.  Meant for small volumes (N <= 128): 8 slots per sample.  `device=False`: round 3's form (triplets from
        the device one projection at a time, scipy sorts and merges on the host) -- kept as the cross-check."""
        from scipy import sparse
        f, be = self._fwd, self.backend
        n_proj = f.poses.shape[0]
        if device and hasattr(be, "csr_assemble") and np.dtype(f.precision) in (np.dtype(np.float32), np.dtype(np.float64)):
            # round 4: everything on the device -- triplets of all projections, mask, sort, duplicate sums, row pointers (csrc/tomo_csr.hip);
            # the host only receives the finished arrays
            data, indices, indptr = be.csr_assemble(f.poses, f._mask, f.precision, max_nnz=max_nnz)      # refuses before anything is downloaded
            idx = np.int32 if max(data.size, f.shape[1]) < 2 ** 31 - 1 else np.int64        # what scipy picks for a matrix of this size
            A = sparse.csr_matrix((data, indices.astype(idx, copy=False), indptr.astype(idx, copy=False)), shape=f.shape)
            return A.T.tocsr() if self._is_adjoint else A
        W, DET, DAT = [], [], []
        total = 0
        for ip in range(n_proj):
            dat, det, wts = be.triplets(f.poses[ip:ip + 1])
            total += dat.size
            if total > max_nnz:
                raise MemoryError("tocsr: more than %d triplets; keep the operator matrix-free" % max_nnz)
            W.append(wts.astype(f.precision, copy=False))                       # :106
            DAT.append(dat)
            DET.append(det.astype(np.int64) + ip * be.n_det)                    # :108
        W, DET, DAT = np.concatenate(W), np.concatenate(DET), np.concatenate(DAT)
        if f._mask is not None:                                                 # :60-70
            m = be.download(f._mask).astype(bool)[DAT]
            if not m.any():
                W = W * 0.0
            else:
                DAT, DET, W = DAT[m], DET[m], W[m]
        A = sparse.csr_matrix(sparse.coo_matrix((W, (DET, DAT)), shape=f.shape))  # :73-76
        return A.T.tocsr() if self._is_adjoint else A


class ProjectionMatrix(object):

    def __init__(self, geometry, precision=np.float32, backend=None):
        self.geometry = geometry
        self.precision = precision
        self.n_proj = None
        self.angles = None
        self.xyz_shift = None
        self.voxel_mask = None
        self._backend = backend
        self._vol_dev = None          # device buffer of the UNPINNED volume last passed to projection_gradient / cost_and_gradient
        self._vol_own = None          # ... the scratch buffer host uploads of unpinned volumes go to
        self._vol_gen = 0             # bumped by every (re)load; memo keys carry it
        self._pinned = None           # the object the caller pinned (identity), see pin_volume()
        self._pin_dev = None          # device buffer holding the pinned volume; never shared with unpinned volumes
        self._pin_own = None          # ... when it was uploaded from a host array
        self._pin_stale = False
        self._last_was_pinned = False  # the volume set_volume() returned last was the pinned one
        self._vol_staged = False      # True while the library's zero-padded copy holds the PINNED volume, unchanged
        self._pg_bufs = None

    @property
    def backend(self):
        if self._backend is None:
            self._backend = HipBackend(self.geometry)     # raises if libtomo_hip.so / GPU is missing
        return self._backend

    def projection_matrix(self, alpha=None, beta=None, phi=None, xyz_shift=None, voxel_mask=None):
        n_proj, phi, alpha, beta, xyz_shift = _normalise_poses(self.geometry, alpha, beta, phi, xyz_shift)
        self.n_proj = n_proj
        self.angles = np.array([phi, alpha, beta]).T
        self.xyz_shift = xyz_shift
        self.voxel_mask = voxel_mask
        cor = np.asarray(self.geometry.cor_shift, np.float64)
        if cor.ndim == 2:
            if cor.shape[0] < n_proj:
                raise ValueError("geometry.cor_shift has %d rows for %d projections" % (cor.shape[0], n_proj))
            cor = cor[:n_proj]      # the reference indexes cor_shift[iproj] (utilities/projection_operators.py:102)
        poses = _lib.poses_array(phi, alpha, beta, xyz_shift, cor)
        return RayOperator(self.backend, poses, self.precision, voxel_mask)

    def _forward_voxel(self):
        """Per-projection triplets of the voxel-driven splat (reference utilities/projection_operators.py:78-93; never
        called there, :54).  Needs projection_matrix() to have set angles / xyz_shift first."""
        import copy
        from . import voxel_utilities
        weights, detector_inds, data_inds = [], [], []
        for iproj in range(self.n_proj):
            this_geo = copy.copy(self.geometry)
            this_geo.cor_shift = np.asarray(self.geometry.cor_shift)[iproj]
            phi, alpha, beta = self.angles[iproj]
            dat, det, wts = voxel_utilities.forward_sparse(this_geo, alpha, beta, phi, self.xyz_shift[iproj], backend=self.backend)
            weights.append(wts.astype(self.precision, copy=False))
            detector_inds.append((det + iproj * int(self.geometry.n_det)).astype(np.int32))
            data_inds.append(dat)
        return weights, detector_inds, data_inds

    def _forward_ray(self):
        """Per-projection triplets of the ray-driven projector (reference utilities/projection_operators.py:95-110: what its
        projection_matrix concatenates into the COO matrix); small volumes only.  Needs projection_matrix() to have set
        angles / xyz_shift first.  `RayOperator.tocsr()` is the assembled form."""
        import copy
        from . import ray_voxel_utilities
        weights, detector_inds, data_inds = [], [], []
        for iproj in range(self.n_proj):
            this_geo = copy.copy(self.geometry)
            this_geo.cor_shift = np.asarray(self.geometry.cor_shift)[iproj]
            phi, alpha, beta = self.angles[iproj]
            dat, det, wts = ray_voxel_utilities.forward_sparse(this_geo, alpha, beta, phi, self.xyz_shift[iproj], backend=self.backend)
            weights.append(wts.astype(self.precision, copy=False))
            data_inds.append(dat.astype(np.int32))
            detector_inds.append(det + iproj * int(self.geometry.n_det))
        return weights, detector_inds, data_inds

    # ---- volume residency for repeated projection_gradient calls (alignment inner loop)
    # The reference recomputes from `rec` on every call (utilities/projection_operators.py:112-122), so by default so does
    # this class: a host `rec` is uploaded and re-staged on EVERY call, a DeviceArray is re-staged on every call.  Keeping a
    # volume resident across calls is explicit and opt-in: `pin_volume(rec)` (or `with P.pinned(rec):`) uploads / stages once
    # and the caller promises not to modify `rec` until `unpin_volume()`; `invalidate_volume()` says "I did modify it".
    # (An earlier version guessed from a fingerprint of a few samples whether a host array had changed; an in-place edit
    # that missed the sampled elements then returned gradients of the old volume.)
    def pin_volume(self, rec):
        """Keep `rec` (host array or DeviceArray) resident in HBM: subsequent calls that pass this very object skip the
        upload and the zero-padded staging.  The caller vouches that the contents do not change until unpin_volume() /
        invalidate_volume()."""
        self._pinned = rec
        return self._load_pinned()

    def unpin_volume(self):
        self._pinned = None
        self._pin_dev = None
        self._pin_stale = False
        self._vol_staged = False

    def invalidate_volume(self):
        """The pinned object's contents have changed: upload / stage again at the next call (it stays pinned)."""
        self._pin_stale = True
        self._vol_staged = False

    def volume_is_pinned(self, rec):
        return self._pinned is not None and rec is self._pinned and not self._pin_stale

    def pinned(self, rec):
        """Context manager form of pin_volume / unpin_volume."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            self.pin_volume(rec)
            try:
                yield self
            finally:
                self.unpin_volume()
        return _cm()

    def _upload_into(self, rec, buf):
        """Device buffer holding `rec`: the DeviceArray itself, or host data uploaded into `buf` (re-allocated on a size change)."""
        be = self.backend
        if be.is_buffer(rec):
            return rec, buf
        flat = np.asarray(rec).reshape(-1)
        if buf is None or buf.size != flat.size:
            buf = be.empty(flat.size)
        buf.upload(flat)
        return buf, buf

    def _load_pinned(self):
        # The pinned volume has a buffer of its own (_pin_own): an unpinned volume passed in between goes to the scratch
        # buffer (_vol_own) and can never overwrite it.  (with one shared buffer, pin(A); call(B); call(A) evaluated
        # A's poses on B's data.)
        self._vol_gen += 1                      # every (re)load is a new generation: nothing derived from the old one is reused
        self._vol_staged = False
        self._pin_stale = False
        self._pin_dev, self._pin_own = self._upload_into(self._pinned, self._pin_own)
        return self._pin_dev

    def _load_volume(self, rec):
        """An UNPINNED volume: uploaded (host array) or taken as it is (DeviceArray) on every call; whatever the library
        has staged for a pinned volume is no longer current afterwards."""
        self._vol_gen += 1
        self._vol_staged = False
        self._vol_dev, self._vol_own = self._upload_into(rec, self._vol_own)
        return self._vol_dev

    def set_volume(self, rec):
        """Device buffer holding `rec` for the next proj_grad / cost_grad call: the pinned copy if `rec` is the pinned
        object, else a fresh upload (host array) / the buffer itself, to be re-staged (DeviceArray)."""
        if self._pinned is not None and rec is self._pinned:
            self._last_was_pinned = True
            return self._load_pinned() if self._pin_stale else self._pin_dev
        self._last_was_pinned = False
        return self._load_volume(rec)

    def pinned_call(self, fn, *args, **kw):
        """Run a proj_grad / cost_grad backend call on the volume set_volume() returned, letting the library reuse its
        staged (zero-padded) copy only while the explicitly pinned volume was also the one staged last and is unchanged."""
        ctx = getattr(self.backend, "ctx", None)
        if ctx is None:
            return fn(*args, **kw)
        pinned_now = self._last_was_pinned and self._pinned is not None and not self._pin_stale
        ctx.set_option("reuse_staged_volume", 1 if (self._vol_staged and pinned_now) else 0)
        self._vol_staged = False
        try:
            out = fn(*args, **kw)
            self._vol_staged = pinned_now        # the library's padded copy now holds the pinned volume -- or another one
        finally:
            ctx.set_option("reuse_staged_volume", 0)
        return out

    def pose_row(self, alpha, beta, phi, xyz_shift, cor_shift):
        return _lib.poses_array([phi], [alpha], [beta], np.asarray(xyz_shift, np.float64).reshape(1, 3),
                                np.asarray(cor_shift, np.float64).reshape(-1)[:3])

    def projection_gradient(self, rec, alpha, beta, phi, xyz_shift, cor_shift):
        be = self.backend
        vol = self.set_volume(rec)
        if self._pg_bufs is None:
            self._pg_bufs = (be.empty(be.n_det), be.empty(6 * be.n_det))
        p_dev, g_dev = self._pg_bufs
        self.pinned_call(be.proj_grad, self.pose_row(alpha, beta, phi, xyz_shift, cor_shift), vol, p_dev, g_dev, 0)
        proj_img = p_dev.download().astype(self.precision, copy=False)
        gradient = g_dev.download().astype(self.precision, copy=False)
        return proj_img.ravel(), gradient.reshape(6, -1)
