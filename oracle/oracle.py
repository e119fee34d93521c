"""
oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy for the small per-projection setup, plain C in
tomo_oracle.c for the ray/sample loops) of the reference's ray-driven
projection hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package
(tomography_alignment_amd) never does.

Parity of this restatement is pinned against outputs of the compiled reference
itself (tests/golden/*.npz made by tests/golden/make_golden.py) in
tests/test_oracle_golden.py.

All citations are path:line in the reference tree (pandekan/tomography_alignment).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtomo_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "libtomo_oracle.so"], stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(path)
        _LIB.orc_trilinear_ray_sparse.restype = ctypes.c_int64
        _LIB.orc_bilinear_sparse.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------------------
# rotations: utilities/rotations.py:9-48
# ----------------------------------------------------------------------------------------
def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([(c, -s, 0.), (s, c, 0.), (0., 0., 1.)])


def der_rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([(-s, -c, 0.), (c, -s, 0.), (0., 0., 0.)])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([(1., 0., 0.), (0., c, -s), (0., s, c)])


def der_rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([(0., 0., 0.), (0., -s, -c), (0., c, -s)])


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([(c, 0., s), (0., 1., 0.), (-s, 0., c)])


def der_rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([(-s, 0., c), (0., 0., 0.), (-c, 0., -s)])


# ----------------------------------------------------------------------------------------
# geometry: utilities/geometry.py:9-105
# ----------------------------------------------------------------------------------------
class Geo(object):
    """Grids and conventions of utilities/geometry.py:14-47,77-105 (restated, not imported)."""

    def __init__(self, n_proj, voxel_shape, voxel_pixsize, detector_shape, detector_pixsize,
                 cor_shift=None, step_size=1.0):
        self.n_proj = n_proj
        self.vox_shape = np.asarray(voxel_shape)
        self.vox_pix = np.asarray(voxel_pixsize, dtype=np.float64)
        self.vox_size = self.vox_shape * self.vox_pix
        self.n_vox = int(np.prod(self.vox_shape))
        self.det_shape = np.asarray(detector_shape)
        self.det_pix = np.asarray(detector_pixsize, dtype=np.float64)
        self.det_size = self.det_shape * self.det_pix
        self.n_det = int(np.prod(self.det_shape))
        self.vox_ds = np.array([1, 1, 1])
        if cor_shift is None:
            self.cor_shift = np.zeros((n_proj, 3))
        else:
            cor_shift = np.asarray(cor_shift, dtype=np.float64)
            self.cor_shift = cor_shift if cor_shift.ndim == 2 else np.tile(cor_shift, n_proj).reshape(n_proj, 3)
        self.step_size = step_size
        nx, ny, nz = self.vox_shape
        sx, sy, sz = self.vox_size
        x = np.linspace(-sx / 2, sx / 2, nx, endpoint=False) + 0.5   # geometry.py:82-84
        y = np.linspace(-sy / 2, sy / 2, ny, endpoint=False) + 0.5
        z = np.linspace(-sz / 2, sz / 2, nz, endpoint=False) + 0.5
        self._axes = (x, y, z)
        self._vc = None                      # (3, n_vox) float64 is 25 GB at 1024^3: built on first use
        self.vox_origin = np.array([x.min(), y.min(), z.min()])
        ndx, ndz = self.det_shape
        dsx, dsz = self.det_size
        xd = np.linspace(-dsx / 2, dsx / 2, ndx, endpoint=False) + 0.5  # geometry.py:92-93
        zd = np.linspace(-dsz / 2, dsz / 2, ndz, endpoint=False) + 0.5
        XD, ZD = np.meshgrid(xd, zd, indexing='ij')
        self.source_centers = np.array([XD.ravel(), -sy * np.ones(self.n_det), ZD.ravel()])
        self.det_centers = np.array([XD.ravel(), sy * np.ones(self.n_det), ZD.ravel()])


    @property
    def vox_centers(self):
        if self._vc is None:
            X, Y, Z = np.meshgrid(*self._axes, indexing='ij')     # geometry.py:85-86
            self._vc = np.array([X.ravel(), Y.ravel(), Z.ravel()])
        return self._vc


# ----------------------------------------------------------------------------------------
# ray setup: utilities/ray_voxel_utilities.py:6-12,53-99
# ----------------------------------------------------------------------------------------
def transform_points(x, alpha, beta, phi, t):
    rot_pa = np.dot(rot_z(phi), rot_x(alpha))               # :8
    xp = np.dot(rot_y(beta), x) + np.asarray(t)[:, np.newaxis]  # :9
    return np.dot(rot_pa, xp)                                   # :10


def ray_setup(geo, alpha, beta, phi, xyz_shift, cor_shift3):
    """utilities/ray_voxel_utilities.py:72-94: returns p0[3,n], rhat[3,n], n_pts, r_len0 and the
    (cor-shifted, untransformed) source/detector points."""
    src = geo.source_centers.copy()
    det = geo.det_centers.copy()
    src[0, :] += cor_shift3[0]                                  # :72-73
    det[0, :] += cor_shift3[0]
    p0 = transform_points(src, alpha, beta, phi, xyz_shift) - geo.vox_origin[:, np.newaxis]
    p1 = transform_points(det, alpha, beta, phi, xyz_shift) - geo.vox_origin[:, np.newaxis]
    r = p1 - p0
    r_length = np.linalg.norm(r, axis=0)
    r_hat = r / r_length
    n = int(r_length[0] / geo.step_size)                        # :88
    return np.ascontiguousarray(p0), np.ascontiguousarray(r_hat), n, float(r_length[0]), src, det


def derivative_ray_points(source_points, ray_vector, alpha, beta, phi, xyz_shift):
    """utilities/ray_voxel_utilities.py:15-50 -> der[9,3,n_rays]."""
    R_p, R_a, R_b = rot_z(phi), rot_x(alpha), rot_y(beta)
    dR_p, dR_a, dR_b = der_rot_z(phi), der_rot_x(alpha), der_rot_y(beta)
    R_pa = np.dot(R_p, R_a)
    R_ab = np.dot(R_a, R_b)
    n = source_points.shape[1]
    der = np.zeros((9, 3, n))
    der[0] = R_pa[:, 0][:, np.newaxis]
    der[1] = R_pa[:, 1][:, np.newaxis]
    der[2] = R_pa[:, 2][:, np.newaxis]
    Rb_st = np.dot(R_b, source_points) + np.asarray(xyz_shift)[:, np.newaxis]
    der[3] = np.dot(dR_p, np.dot(R_a, Rb_st))
    der[4] = np.dot(R_p, np.dot(dR_a, Rb_st))
    der[5] = np.dot(R_pa, np.dot(dR_b, source_points))
    der[6] = np.dot(dR_p, np.dot(R_ab, ray_vector))[:, np.newaxis]
    der[7] = np.dot(R_p, np.dot(dR_a, np.dot(R_b, ray_vector)))[:, np.newaxis]
    der[8] = np.dot(R_pa, np.dot(dR_b, ray_vector))[:, np.newaxis]
    return der


# ----------------------------------------------------------------------------------------
# A1+A3: forward_sparse -> triplets ; projection_matrix -> CSR
# ----------------------------------------------------------------------------------------
def forward_sparse(geo, alpha, beta, phi, xyz_shift, cor_shift3):
    """utilities/ray_voxel_utilities.py:53-110 + src/ray_wt_grad.f90:1-92."""
    p0, rhat, n, _, _, _ = ray_setup(geo, alpha, beta, phi, xyz_shift, cor_shift3)
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    cap = 8 * geo.n_det * n
    dat = np.empty(cap, np.int32)
    det = np.empty(cap, np.int32)
    wts = np.empty(cap, np.float64)
    k = _lib().orc_trilinear_ray_sparse(_p(p0), _p(rhat), ctypes.c_int64(geo.n_det), n,
                                        ctypes.c_double(geo.step_size), nx, ny, nz,
                                        _p(dat), _p(det), _p(wts))
    return dat[:k], det[:k], wts[:k]


def _default_poses(geo, alpha, beta, phi, xyz_shift):
    """utilities/projection_operators.py:24-52 defaults and n_proj==1 re-wrapping."""
    if phi is None:
        n_proj = geo.n_proj
        phi = np.linspace(0., np.pi, n_proj)
    else:
        n_proj = np.size(phi)
    alpha = np.zeros_like(phi) if alpha is None else alpha
    beta = np.zeros_like(phi) if beta is None else beta
    xyz_shift = np.zeros((n_proj, 3)) if xyz_shift is None else xyz_shift
    phi = np.atleast_1d(np.squeeze(phi)).astype(np.float64)
    alpha = np.atleast_1d(np.squeeze(alpha)).astype(np.float64)
    beta = np.atleast_1d(np.squeeze(beta)).astype(np.float64)
    xyz_shift = np.asarray(xyz_shift, dtype=np.float64).reshape(n_proj, 3)
    return n_proj, alpha, beta, phi, xyz_shift


def projection_matrix(geo, alpha=None, beta=None, phi=None, xyz_shift=None, voxel_mask=None,
                      precision=np.float32):
    """utilities/projection_operators.py:22-76 -> scipy CSR (duplicates summed, zeros kept)."""
    from scipy import sparse
    n_proj, alpha, beta, phi, xyz_shift = _default_poses(geo, alpha, beta, phi, xyz_shift)
    W, DET, DAT = [], [], []
    for ip in range(n_proj):
        d, r, w = forward_sparse(geo, alpha[ip], beta[ip], phi[ip], xyz_shift[ip], geo.cor_shift[ip])
        W.append(w.astype(precision, copy=False))
        DAT.append(d.astype(np.int32))
        DET.append(r + ip * geo.n_det)
    W, DET, DAT = np.concatenate(W), np.concatenate(DET), np.concatenate(DAT)
    if voxel_mask is not None:
        vm = voxel_mask.ravel().astype(bool)
        m = vm[DAT]
        if np.sum(m) == 0:
            W *= 0.0
        else:
            DAT, DET, W = DAT[m], DET[m], W[m]
    A = sparse.coo_matrix((W, (DET, DAT)), shape=(n_proj * geo.n_det, geo.n_vox))
    return sparse.csr_matrix(A)


# ----------------------------------------------------------------------------------------
# matrix-free A.x and A^T.y with the assembled operator's semantics (recon/sirt.py:59,61)
# ----------------------------------------------------------------------------------------
def forward(geo, rec, alpha=None, beta=None, phi=None, xyz_shift=None):
    n_proj, alpha, beta, phi, xyz_shift = _default_poses(geo, alpha, beta, phi, xyz_shift)
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    rec = np.ascontiguousarray(rec, dtype=np.float32).ravel()
    out = np.zeros((n_proj, geo.n_det), np.float64)
    for ip in range(n_proj):
        p0, rhat, n, _, _, _ = ray_setup(geo, alpha[ip], beta[ip], phi[ip], xyz_shift[ip], geo.cor_shift[ip])
        _lib().orc_forward(_p(p0), _p(rhat), ctypes.c_int64(geo.n_det), n, ctypes.c_double(geo.step_size),
                           nx, ny, nz, _p(rec), _p(out[ip]))
    return out


def adjoint(geo, y, alpha=None, beta=None, phi=None, xyz_shift=None, coloured_rows=False):
    """coloured_rows=True: the atomic-free multi-thread form (orc_adjoint_rows; detector pitch >= voxel only) -- used by
    bench.py's all-cores timing, same sums."""
    n_proj, alpha, beta, phi, xyz_shift = _default_poses(geo, alpha, beta, phi, xyz_shift)
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    y = np.ascontiguousarray(y, dtype=np.float32).reshape(n_proj, geo.n_det)
    vol = np.zeros(geo.n_vox, np.float64)
    for ip in range(n_proj):
        p0, rhat, n, _, _, _ = ray_setup(geo, alpha[ip], beta[ip], phi[ip], xyz_shift[ip], geo.cor_shift[ip])
        if coloured_rows:
            _lib().orc_adjoint_rows(_p(p0), _p(rhat), ctypes.c_int64(geo.n_det), ctypes.c_int64(int(geo.det_shape[1])), n,
                                    ctypes.c_double(geo.step_size), nx, ny, nz, _p(y[ip]), _p(vol))
        else:
            _lib().orc_adjoint(_p(p0), _p(rhat), ctypes.c_int64(geo.n_det), n, ctypes.c_double(geo.step_size),
                               nx, ny, nz, _p(y[ip]), _p(vol))
    return vol


def _ray_subset(geo, phi, ray_index, alpha=0.0, beta=0.0, xyz_shift=None):
    p0, rhat, n, _, _, _ = ray_setup(geo, alpha, beta, phi, np.zeros(3) if xyz_shift is None else xyz_shift, geo.cor_shift[0])
    idx = np.asarray(ray_index, np.int64)
    return np.ascontiguousarray(p0[:, idx]), np.ascontiguousarray(rhat[:, idx]), n, idx


def forward_rays(geo, rec, phi, ray_index, **kw):
    """A x restricted to the rays `ray_index` of one projection (same loop as forward(); bench.py times a fraction of an
    angle's rays on one thread with it)."""
    p0, rhat, n, idx = _ray_subset(geo, phi, ray_index, **kw)
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    rec = np.ascontiguousarray(rec, dtype=np.float32).ravel()
    out = np.zeros(idx.size, np.float64)
    _lib().orc_forward(_p(p0), _p(rhat), ctypes.c_int64(idx.size), n, ctypes.c_double(geo.step_size), nx, ny, nz, _p(rec), _p(out))
    return out


def adjoint_rays(geo, y, phi, ray_index, **kw):
    """A^T y restricted to the rays `ray_index` of one projection (y holds their values)."""
    p0, rhat, n, idx = _ray_subset(geo, phi, ray_index, **kw)
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    y = np.ascontiguousarray(y, dtype=np.float32).ravel()
    vol = np.zeros(geo.n_vox, np.float64)
    _lib().orc_adjoint(_p(p0), _p(rhat), ctypes.c_int64(idx.size), n, ctypes.c_double(geo.step_size), nx, ny, nz, _p(y), _p(vol))
    return vol


# ----------------------------------------------------------------------------------------
# A2+A4: projection + 6-DoF gradient (utilities/projection_operators.py:112-122)
# ----------------------------------------------------------------------------------------
def projection_gradient(geo, rec, alpha, beta, phi, xyz_shift, cor_shift3, precision=np.float32):
    """utilities/ray_voxel_utilities.py:113-170 + src/ray_wt_grad.f90:95-223.
    Returns (proj[n_det], grad[6,n_det]) rows tx,ty,tz,phi,alpha,beta."""
    p0, rhat, n, r_len0, src, det = ray_setup(geo, alpha, beta, phi, xyz_shift, cor_shift3)
    ray_vec = (det - src)[:, 0]
    der = np.ascontiguousarray(derivative_ray_points(src, ray_vec, alpha, beta, phi, xyz_shift))
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    rec64 = np.ascontiguousarray(np.asarray(rec).ravel(), dtype=np.float64)
    img = np.zeros(geo.n_det, np.float64)
    grad = np.zeros((6, geo.n_det), np.float64)
    _lib().orc_trilinear_ray_interp(_p(p0), _p(rhat), ctypes.c_int64(geo.n_det), n,
                                    ctypes.c_double(geo.step_size), ctypes.c_double(r_len0),
                                    nx, ny, nz, _p(rec64), _p(der), _p(img), _p(grad))
    return img.astype(precision, copy=False), grad.astype(precision, copy=False)


def projection_gradient_rays(geo, rec64, alpha, beta, phi, xyz_shift, cor_shift3, ray_index):
    """projection_gradient restricted to the rays `ray_index` (same loop; bench.py times a fraction of a pose's rays with it).
    `rec64` must already be a contiguous float64 array (no per-call conversion of a large volume)."""
    p0, rhat, n, r_len0, src, det = ray_setup(geo, alpha, beta, phi, xyz_shift, cor_shift3)
    idx = np.asarray(ray_index, np.int64)
    ray_vec = (det - src)[:, 0]
    der = np.ascontiguousarray(derivative_ray_points(src[:, idx], ray_vec, alpha, beta, phi, xyz_shift))
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    assert rec64.dtype == np.float64 and rec64.flags["C_CONTIGUOUS"]
    img = np.zeros(idx.size, np.float64)
    grad = np.zeros((6, idx.size), np.float64)
    _lib().orc_trilinear_ray_interp(_p(np.ascontiguousarray(p0[:, idx])), _p(np.ascontiguousarray(rhat[:, idx])), ctypes.c_int64(idx.size), n,
                                    ctypes.c_double(geo.step_size), ctypes.c_double(r_len0),
                                    nx, ny, nz, _p(rec64.ravel()), _p(der), _p(img), _p(grad))
    return img, grad


def ray_face_distance(geo, alpha, beta, phi, xyz_shift, cor_shift3):
    """Test aid: per ray, the smallest distance (voxels) of an in-volume sample to a cell face (orc_ray_face_distance).
    The pose gradient of a ray with a sample within position rounding of a face is ill-conditioned (the interpolant's
    spatial gradient jumps there, its value does not)."""
    p0, rhat, n, r_len0, src, det = ray_setup(geo, alpha, beta, phi, xyz_shift, cor_shift3)
    nx, ny, nz = (int(v) for v in geo.vox_shape)
    out = np.ones(geo.n_det, np.float64)
    _lib().orc_ray_face_distance(_p(p0), _p(rhat), ctypes.c_int64(geo.n_det), n, ctypes.c_double(geo.step_size),
                                 nx, ny, nz, _p(out))
    return out


def set_threads(n):
    """OpenMP threads of the ray loops (forward, adjoint, projection_gradient); 1 = the serial reference's shape."""
    _lib().orc_set_threads(int(n))


# ----------------------------------------------------------------------------------------
# A6: voxel-driven back_project (src/back_projection.f90:1-34), float32
# ----------------------------------------------------------------------------------------
def back_project_voxel(geo, det, alpha, beta, phi, xyz):
    """det: [n_proj, ndx, ndz] float32 C-order.  Origin = vox_origin (the detector grid of the
    matrix-free routine shares the voxel origin in x,z when detector == volume footprint)."""
    n_proj = np.size(phi)
    ndx, ndz = (int(v) for v in geo.det_shape)
    alpha = np.ascontiguousarray(alpha, np.float32)
    beta = np.ascontiguousarray(beta, np.float32)
    phi = np.ascontiguousarray(phi, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(n_proj, 3)
    vc = np.ascontiguousarray(geo.vox_centers, np.float32)
    org = np.ascontiguousarray(geo.vox_origin, np.float32)
    det = np.ascontiguousarray(det, np.float32).reshape(n_proj, ndx, ndz)
    atx = np.zeros(geo.n_vox, np.float32)
    _lib().orc_back_project_voxel(_p(alpha), _p(beta), _p(phi), _p(xyz), _p(vc), _p(org), _p(det),
                                  n_proj, ctypes.c_int64(geo.n_vox), ndx, ndz, _p(atx))
    return atx


# ----------------------------------------------------------------------------------------
# A9: voxel-driven splat (utilities/voxel_utilities.py:6-108 + src/vox_wt_grad.f90)
# ----------------------------------------------------------------------------------------
def _vox_rigid(x, alpha, beta, phi, xyz):
    rtx = np.dot(rot_z(phi), x)                              # voxel_utilities.py:16-18
    ratx = np.dot(rot_x(alpha), rtx)
    return np.dot(rot_y(beta), ratx + np.asarray(xyz)[:, np.newaxis])


def _vox_floor_alpha(geo, alpha, beta, phi, xyz_shift, cor_shift3):
    rc = _vox_rigid(geo.vox_centers, alpha, beta, phi, xyz_shift)
    orig = geo.vox_origin - cor_shift3                       # voxel_utilities.py:61
    dx = geo.vox_ds
    fx = np.floor((rc[0] - orig[0]) / dx[0]).astype(np.int32)
    fz = np.floor((rc[2] - orig[2]) / dx[2]).astype(np.int32)
    ax = ((rc[0] - orig[0] - fx * dx[0]) / dx[0]).astype(np.float32)
    az = ((rc[2] - orig[2] - fz * dx[2]) / dx[2]).astype(np.float32)
    return fx, fz, ax, az


def vox_forward_sparse(geo, alpha, beta, phi, xyz_shift, cor_shift3):
    fx, fz, ax, az = _vox_floor_alpha(geo, alpha, beta, phi, xyz_shift, cor_shift3)
    cap = 4 * geo.n_vox
    dat = np.empty(cap, np.int32)
    det = np.empty(cap, np.int32)
    wts = np.empty(cap, np.float32)
    k = _lib().orc_bilinear_sparse(ctypes.c_int64(geo.n_vox), _p(fx), _p(fz), _p(ax), _p(az),
                                   int(geo.det_shape[0]), int(geo.det_shape[1]), _p(dat), _p(det), _p(wts))
    return dat[:k], det[:k], wts[:k]


def vox_derivative_rigid(x, a, b, t, s):
    """utilities/voxel_utilities.py:23-48."""
    R_b, R_a, R_t = rot_y(b), rot_x(a), rot_z(t)
    dR_b, dR_a, dR_t = der_rot_y(b), der_rot_x(a), der_rot_z(t)
    rtx = np.dot(R_t, x)
    ratx = np.dot(R_a, rtx)
    rba = np.dot(R_b, R_a)
    der = np.zeros((6, x.shape[0], x.shape[1]))
    der[0] = R_b[:, 0][:, np.newaxis]
    der[1] = R_b[:, 1][:, np.newaxis]
    der[2] = R_b[:, 2][:, np.newaxis]
    der[3] = np.dot(rba, np.dot(dR_t, x))
    der[4] = np.dot(R_b, np.dot(dR_a, rtx))
    der[5] = np.dot(dR_b, ratx + np.asarray(s)[:, np.newaxis])
    return der


def vox_forward_proj_grad(geo, alpha, beta, phi, xyz_shift, cor_shift3, rec):
    """utilities/voxel_utilities.py:82-108 -> (det_img.ravel(), gradient.reshape(6,-1))."""
    fx, fz, ax, az = _vox_floor_alpha(geo, alpha, beta, phi, xyz_shift, cor_shift3)
    der = np.ascontiguousarray(vox_derivative_rigid(geo.vox_centers, alpha, beta, phi, xyz_shift), np.float32)
    ndx, ndz = int(geo.det_shape[0]), int(geo.det_shape[1])
    rec32 = np.ascontiguousarray(np.asarray(rec).ravel(), np.float32)
    img = np.zeros(ndx * ndz, np.float32)
    grad = np.zeros(6 * ndx * ndz, np.float32)
    _lib().orc_bilinear_vox_interp(ctypes.c_int64(geo.n_vox), _p(fx), _p(fz), _p(ax), _p(az), _p(rec32),
                                   ndx, ndz, _p(der), _p(img), _p(grad))
    img = np.reshape(img, (ndz, ndx), order='F')
    grad = np.reshape(grad, (6, ndz, ndx), order='F')
    return img.ravel(), grad.reshape(6, -1)


def bilinear_sparse(n_vox, floor_x, floor_z, alpha_x, alpha_z, ndim_x, ndim_z):
    """src/vox_wt_grad.f90:58-112 with the f2py module's signature and returns: (dat_inds, det_inds, wts, n_inds), length 4*n_vox,
    -999 beyond n_inds (:73-75).  Pinned bit for bit to the f2py module by tests/golden/g13."""
    n = int(n_vox)
    fx, fz = np.ascontiguousarray(floor_x, np.int32), np.ascontiguousarray(floor_z, np.int32)
    ax, az = np.ascontiguousarray(alpha_x, np.float32), np.ascontiguousarray(alpha_z, np.float32)
    dat, det, wts = np.full(4 * n, -999, np.int32), np.full(4 * n, -999, np.int32), np.full(4 * n, -999.0, np.float32)
    k = _lib().orc_bilinear_sparse(ctypes.c_int64(n), _p(fx), _p(fz), _p(ax), _p(az), int(ndim_x), int(ndim_z), _p(dat), _p(det), _p(wts))
    return dat, det, wts, int(k)


def bilinear_vox_interp(n_vox, floor_x, floor_z, alpha_x, alpha_z, rec, ndim_x, ndim_z, der_points):
    """src/vox_wt_grad.f90:1-55 with the f2py module's signature and returns: det_img (ndim_z, ndim_x), grad_det_img (6, ndim_z, ndim_x),
    float32, Fortran order.  Pinned bit for bit to the f2py module by tests/golden/g13."""
    n, ndx, ndz = int(n_vox), int(ndim_x), int(ndim_z)
    fx, fz = np.ascontiguousarray(floor_x, np.int32), np.ascontiguousarray(floor_z, np.int32)
    ax, az = np.ascontiguousarray(alpha_x, np.float32), np.ascontiguousarray(alpha_z, np.float32)
    rec32 = np.ascontiguousarray(np.asarray(rec).ravel(), np.float32)
    der = np.ascontiguousarray(der_points, np.float32)               # C-order (6, 3, n_vox): what orc_bilinear_vox_interp indexes
    img, grad = np.zeros(ndx * ndz, np.float32), np.zeros(6 * ndx * ndz, np.float32)
    _lib().orc_bilinear_vox_interp(ctypes.c_int64(n), _p(fx), _p(fz), _p(ax), _p(az), _p(rec32), ndx, ndz, _p(der), _p(img), _p(grad))
    return np.reshape(img, (ndz, ndx), order='F'), np.reshape(grad, (6, ndz, ndx), order='F')


# ----------------------------------------------------------------------------------------
# solvers restated on top of forward/adjoint callables (recon/sirt.py, recon/cgls.py)
# ----------------------------------------------------------------------------------------
def sirt(fwd, adj, n_vox, projections, niter, positivity=False, ground_truth=None, rec=None,
         stop_rule_k=0, zero_guard=None):
    """recon/sirt.py:26-107.  fwd(x)->[n_proj*n_det], adj(y)->[n_vox] (float32 in/out).
    stop_rule_k=0 / zero_guard=None restate recon/sirt.py:37-38,75;  stop_rule_k=1 /
    zero_guard=1e-8 restate recon/sirt_mpi.py:69-70,116."""
    b = np.asarray(projections, np.float32)
    n_rows = b.size
    W = np.asarray(fwd(np.ones(n_vox, np.float32)), np.float32).ravel()
    V = np.asarray(adj(np.ones(n_rows, np.float32)), np.float32).ravel()
    if zero_guard is None:
        V[V == 0.] = np.inf
        W[W == 0.] = np.inf
    else:
        V[V < zero_guard] = np.inf
        W[W < zero_guard] = np.inf
    V = (1. / V).astype(np.float32)
    W = (1. / W).astype(np.float32)
    rec = np.zeros(n_vox, np.float32) if rec is None else np.array(rec, np.float32).ravel()
    if ground_truth is not None:
        gt = np.asarray(ground_truth).ravel()
        norm_factor = np.linalg.norm(gt)
    else:
        norm_factor = np.linalg.norm(b)
    rms = np.zeros(niter)
    k, stop = 0, 0
    while k < niter and not stop:
        res = b.ravel() - np.asarray(fwd(rec), np.float32).ravel()
        bp = np.asarray(adj((W * res).astype(np.float32)), np.float32).ravel()
        rec += bp * V
        if positivity:
            rec[rec < 0.] = 0.
        conv = np.linalg.norm(res)
        rms[k] = conv / norm_factor if ground_truth is None else np.linalg.norm(gt - rec) / norm_factor
        if k > stop_rule_k and rms[k] > rms[k - 1]:
            stop = 1
        k += 1
    return rec, rms[:k]


class Cgls(object):
    """recon/cgls.py:7-104, the sparse-matrix branch (`self.method` is undefined in the reference so only the csr path can be meant,
    :51-54), as a stateful object like the reference's class: `fwd` / `adj` may be replaced between two `run` calls while `_r`, `_p`,
    `_gamma` stay (what golden G11 c / d do to the reference to make its re-initialisation rule fire, :60-68).
    Pinned by tests/test_oracle_golden.py::test_g11_cgls_restatement_vs_reference_class."""

    def __init__(self, fwd, adj, n_vox, projections, ground_truth=None, rec=None):
        self.fwd, self.adj = fwd, adj
        self.b = np.asarray(projections, np.float32).ravel()
        self.rec = np.zeros(n_vox, np.float32) if rec is None else np.array(rec, np.float32).ravel()
        self.ground_truth = ground_truth
        self.reinit_lines, self.quit = 0, False
        self._init()

    def _init(self):                                                    # _initialize, :26-36
        self._r = self.b - np.asarray(self.fwd(self.rec), np.float32).ravel()
        self._p = np.asarray(self.adj(self._r), np.float32).ravel()
        self._gamma = np.linalg.norm(self._p) ** 2

    def run(self, niter):                                               # run_main_iteration, :38-104
        b, rec = self.b, self.rec
        norm_factor = np.linalg.norm(b) if self.ground_truth is None else np.linalg.norm(self.ground_truth)
        rms = np.zeros(niter)
        conv = np.zeros(niter)
        k, reinit_iter = 0, 0
        while k < niter:
            r = np.asarray(self.fwd(self._p), np.float32).ravel()       # :54
            a = self._gamma / np.linalg.norm(r) ** 2                    # :56
            rec += (a * self._p).astype(np.float32)                     # :57
            conv[k] = np.linalg.norm(b - np.asarray(self.fwd(rec), np.float32).ravel())     # :58-59
            if k > 0 and conv[k] > conv[k - 1]:                         # :60
                self.reinit_lines += 1
                if reinit_iter + 1 == k:                                # :63 (reinit_iter starts at 0: a rise at k = 1 quits)
                    self.quit = True
                    return rec, rms[:k]
                rec -= (a * self._p).astype(np.float32)                 # :66
                self._init()                                            # :67
                reinit_iter = k
            self._r = self._r - (a * r).astype(np.float32)              # :70 (after a re-initialisation: the step and r of before it)
            p = np.asarray(self.adj(self._r), np.float32).ravel()       # :72
            gamma = np.linalg.norm(p) ** 2
            beta = gamma / self._gamma
            self._gamma = gamma
            self._p = (p + beta * self._p).astype(np.float32)           # :78
            rms[k] = (np.linalg.norm(self._r) / norm_factor if self.ground_truth is None
                      else np.linalg.norm(rec - np.asarray(self.ground_truth).ravel()) / norm_factor)   # :79-82
            k += 1
        return rec, rms[:k]


def cgls(fwd, adj, n_vox, projections, niter, ground_truth=None, rec=None):
    """One run of `Cgls` from a fresh state (recon/cgls.py:26-104)."""
    return Cgls(fwd, adj, n_vox, projections, ground_truth=ground_truth, rec=rec).run(niter)


# ----------------------------------------------------------------------------------------
# regularised solvers' vector kernels: recon/regularized.py:433-440, utilities/tv_denoise.py
# ----------------------------------------------------------------------------------------
def soft_thresholding(x, lam):
    """recon/regularized.py:433-440: shrink towards zero by lam (zero inside [-lam, lam])."""
    x = np.asarray(x)
    out = np.zeros_like(x)
    up, dn = x > lam, x < -lam
    out[up] = x[up] - lam
    out[dn] = x[dn] + lam
    return out


def tv_gradient(img):
    """utilities/tv_denoise.py:34-59: forward differences along every axis, zero at an axis' last index."""
    g = np.zeros((img.ndim,) + img.shape, dtype=img.dtype)
    for d in range(img.ndim):
        dst = [slice(None)] * img.ndim
        dst[d] = slice(None, -1)
        g[d][tuple(dst)] = np.diff(img, axis=d)
    return g


def tv_div(grad):
    """utilities/tv_denoise.py:20-31: negative adjoint of tv_gradient, accumulated axis by axis in the reference's order
    (res[:-1] += g[:-1]; res[1:-1] -= g[:-2]; res[-1] -= g[-2])."""
    res = np.zeros(grad.shape[1:], dtype=grad.dtype)
    for d in range(grad.shape[0]):
        g = np.moveaxis(grad[d], d, 0)
        r = np.moveaxis(res, d, 0)           # a view: the in-place updates land in res
        r[:-1] += g[:-1]
        r[1:-1] -= g[:-2]
        r[-1] -= g[-2]
    return res


def tv_norm_3d(x):
    """utilities/tv_denoise.py:62-64."""
    return np.linalg.norm(tv_gradient(x))


def tv_dual_gap(im, new, gap, weight):
    """utilities/tv_denoise.py:78-95 (3-D branch)."""
    im_norm = (im ** 2).sum()
    g = tv_gradient(new)
    tv_new = 2 * weight * np.sqrt(g[0] ** 2 + g[1] ** 2 + g[2] ** 2).sum()
    return 0.5 / im_norm * ((gap ** 2).sum() + tv_new - im_norm + (new ** 2).sum())


def tv_denoise_fista(im, weight=50, niter=200, eps=1.e-5, check_gap_frequency=3, return_info=False):
    """utilities/tv_denoise.py:98-170 for a 3-D volume: FISTA on the dual of 0.5*||im - res||^2 + weight*TV(res).  Returns
    `new` as the reference does -- the iterate of the LAST dual-gap check (every check_gap_frequency iterations)."""
    im = np.asarray(im)
    assert im.ndim == 3
    factor = 12.0                                                      # :139-142
    grad_im = np.zeros((3,) + im.shape, dtype=im.dtype)
    grad_aux = np.zeros((3,) + im.shape, dtype=im.dtype)
    t, i, dgap = 1., 0, 0.0
    new = im.copy()
    while i < niter:
        error = weight * tv_div(grad_aux) - im                         # :151
        grad_tmp = tv_gradient(error)
        grad_tmp *= 1 / (factor * weight)                              # :153
        grad_aux += grad_tmp
        norm = np.maximum(np.sqrt(np.sum(grad_aux ** 2, 0)), 1.)       # :67-75 projection on the unit ball, in place
        for comp in grad_aux:
            comp /= norm
        grad_tmp = grad_aux
        t_new = 0.5 * (1 + np.sqrt(1 + 4 * t ** 2))
        t_factor = (t - 1) / t_new
        grad_aux = (1 + t_factor) * grad_tmp - t_factor * grad_im      # :158
        grad_im = grad_tmp
        t = t_new
        if (i % check_gap_frequency) == 0:                             # :161-166
            gap = weight * tv_div(grad_im)
            new = im - gap
            dgap = tv_dual_gap(im, new, gap, weight)
            if dgap < eps:
                break
        i += 1
    return (new, i, float(dgap)) if return_info else new


# ----------------------------------------------------------------------------------------
# phantom (utilities/generate_phantom.py:28-46,81-109,194-209 -- tomopy-derived Shepp-Logan)
# ----------------------------------------------------------------------------------------
_SHEPP = [  # A, a, b, c, x0, y0, z0, phi, theta, psi   (generate_phantom.py:194-209)
    [1., .6900, .920, .810, 0., 0., 0., 90., 90., 90.],
    [-.8, .6624, .874, .780, 0., -.0184, 0., 90., 90., 90.],
    [-.2, .1100, .310, .220, .22, 0., 0., -108., 90., 100.],
    [-.2, .1600, .410, .280, -.22, 0., 0., 108., 90., 100.],
    [.1, .2100, .250, .410, 0., .35, -.15, 90., 90., 90.],
    [.1, .0460, .046, .050, 0., .1, .25, 90., 90., 90.],
    [.1, .0460, .046, .050, 0., -.1, .25, 90., 90., 90.],
    [.1, .0460, .023, .050, -.08, -.605, 0., 90., 90., 90.],
    [.1, .0230, .023, .020, 0., -.606, 0., 90., 90., 90.],
    [.1, .0230, .046, .020, .06, -.605, 0., 90., 90., 90.]]


def shepp3d(size):
    """Restated from the tomopy-derived generator; pinned by golden G7 (the table above is
    verified entry-by-entry against the golden's recorded parameter array in the test)."""
    size = (size, size, size) if np.isscalar(size) else tuple(size)
    obj = np.zeros(size, dtype=np.float32)
    rng = [np.linspace(-1, 1, n) for n in size]
    X, Y, Z = np.meshgrid(*rng, indexing='ij')
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()])
    for A, a, b, c, x0, y0, z0, phi, theta, psi in _SHEPP:
        p, t, s = np.radians(phi), np.radians(theta), np.radians(psi)
        cp, sp, ct, st, cs, ss = np.cos(p), np.sin(p), np.cos(t), np.sin(t), np.cos(s), np.sin(s)
        R = np.array([[cs * cp - ct * sp * ss, cs * sp + ct * cp * ss, ss * st],
                      [-ss * cp - ct * sp * cs, -ss * sp + ct * cp * cs, cs * st],
                      [st * sp, -st * cp, ct]])
        q = (np.dot(R, coords) - np.array([[x0], [y0], [z0]])) / np.array([[a], [b], [c]])
        inside = np.square(q).sum(axis=0) <= 1.0  # rotate, shift, scale: generate_phantom.py:170-179
        m = inside.reshape(size)
        obj[m] = (obj[m].astype(np.float64) + A).astype(np.float32)  # f64 add, f32 store (:131)
    return obj.clip(0, np.inf)
