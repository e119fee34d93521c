"""
`src.ray_wt_grad` -- the reference's f2py module (src/ray_wt_grad.f90), same two functions, same call signatures, same
returns, running on libtomo_hip.so (tomo_trilinear_ray_sparse / tomo_trilinear_ray_interp, csrc/tomo_f2py.hip).

This is the LOWEST binding level: the reference's own utilities/ray_voxel_utilities.py builds the (3, n_rays, n_points)
sample tables in numpy (:85-99,151) and calls

    dat_inds, det_inds, wts, n_inds = ray_wt_grad.trilinear_ray_sparse(floor_points, w_floor, nx, ny, nz, n_rays, n_points)   (:103)
    det_img, grad_det_img = ray_wt_grad.trilinear_ray_interp(floor_points, w_floor, nx, ny, nz, n_rays, n_points, rec, step, der)   (:164)

With this module in place of the extension those lines run unedited.  It is a compatibility surface: the tables are 36
bytes per sample and travel over PCIe on every call; the operator API (utilities/projection_operators.py) is the fast path.
Arrays are converted as f2py converts them (Fortran order, int32 / float64); `intent(out)` arrays are allocated here.
There is no CPU fallback: without the library or a GPU the first call raises TomoError.
"""
import ctypes

import numpy as np

try:
    from .. import _lib
except ImportError:      # the package directory itself on sys.path (`import src`)
    import _lib

_ctx = None


def _context():
    global _ctx
    if _ctx is None or _ctx._h is None:      # closed behind our back (a test harness closing what a test opened): open another
        _ctx = _lib.Context()
    return _ctx


def _f(a, dtype):
    return np.asfortranarray(a, dtype=dtype)


def trilinear_ray_sparse(floor_points, w_floor, nx, ny, nz, n_rays, n_points):
    """src/ray_wt_grad.f90:1-92.  Returns (dat_inds, det_inds, wts, n_inds); the arrays have length 8*n_rays*n_points and are
    -999 beyond n_inds, as the Fortran leaves them."""
    fp, wf = _f(floor_points, np.int32), _f(w_floor, np.float64)
    n_rays, n_points = int(n_rays), int(n_points)
    if fp.shape != (3, n_rays, n_points) or wf.shape != (3, n_rays, n_points):
        raise ValueError("floor_points / w_floor must have shape (3, n_rays, n_points)")
    m = 8 * n_rays * n_points
    dat, det, wts = np.empty(m, np.int32), np.empty(m, np.int32), np.empty(m, np.float64)
    n = ctypes.c_int32(0)
    c = _context()
    c.check(c.lib.tomo_trilinear_ray_sparse(c.handle, fp.ctypes.data_as(ctypes.c_void_p), _lib.dptr(wf), int(nx), int(ny), int(nz), n_rays, n_points,
                                            dat.ctypes.data_as(ctypes.c_void_p), det.ctypes.data_as(ctypes.c_void_p), _lib.dptr(wts), ctypes.byref(n)))
    return dat, det, wts, int(n.value)


def trilinear_ray_interp(floor_points, w_floor, nx, ny, nz, n_rays, n_points, recon, step, der):
    """src/ray_wt_grad.f90:95-223.  Returns (det_img [n_rays], grad_det_img [6, n_rays]) in float64, rows tx, ty, tz, phi, alpha, beta."""
    fp, wf = _f(floor_points, np.int32), _f(w_floor, np.float64)
    n_rays, n_points = int(n_rays), int(n_points)
    rec = np.ascontiguousarray(recon, np.float64).ravel()
    st, dr = _f(step, np.float64), _f(der, np.float64)
    if fp.shape != (3, n_rays, n_points) or wf.shape != (3, n_rays, n_points) or st.shape != (n_rays, n_points) or dr.shape != (9, 3, n_rays):
        raise ValueError("shapes: floor_points, w_floor (3, n_rays, n_points); step (n_rays, n_points); der (9, 3, n_rays)")
    if rec.size < int(nx) * int(ny) * int(nz):
        raise ValueError("recon is shorter than nx*ny*nz")
    img = np.empty(n_rays, np.float64)
    grad = np.empty((6, n_rays), np.float64, order="F")
    c = _context()
    c.check(c.lib.tomo_trilinear_ray_interp(c.handle, fp.ctypes.data_as(ctypes.c_void_p), _lib.dptr(wf), int(nx), int(ny), int(nz), n_rays, n_points,
                                            _lib.dptr(rec), _lib.dptr(st), _lib.dptr(dr), _lib.dptr(img), _lib.dptr(grad)))
    return img, grad
