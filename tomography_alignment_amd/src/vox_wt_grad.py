"""
`src.vox_wt_grad` -- the reference's OTHER f2py module (src/vox_wt_grad.f90), same two functions, same call signatures, same
returns, running on libtomo_hip.so (tomo_bilinear_sparse / tomo_bilinear_vox_interp, csrc/tomo_f2py.hip).

The reference's utilities/voxel_utilities.py builds floor_x / floor_z / alpha_x / alpha_z (and the (6, 3, n_vox) derivative table) in
numpy (:59-67,88-96) and calls

    dat_inds, det_inds, wts, n_inds = vox_wt_grad.bilinear_sparse(n_vox, floor_x, floor_z, alpha_x, alpha_z, ndim_x, ndim_z)                 (:69)
    det_img, grad_det_img = vox_wt_grad.bilinear_vox_interp(n_vox, floor_x, floor_z, alpha_x, alpha_z, rec, ndim_x, ndim_z, der_points)      (:98)

`from src import vox_wt_grad` (utilities/voxel_utilities.py:3) resolves here with the package directory ahead of the reference's `src/`
on sys.path -- which is also what lets the reference's utilities/projection_operators.py (:7-8 imports both utilities modules) be
imported at all without the compiled extensions.  Arrays are converted as f2py converts them (int32 / float32, Fortran order for
der_points); the `intent(out)` arrays are allocated here and returned with the shapes and memory order f2py returns: det_img
(ndim_z, ndim_x) and grad_det_img (6, ndim_z, ndim_x), Fortran-contiguous float32 -- the callers' `.ravel()` / `.reshape(6, -1)` depend on it.
Single precision in the reference's operation order, additions into a pixel in voxel order: bit-identical to the f2py module
(tests/golden/g13).  There is no CPU fallback: without the library or a GPU the first call raises TomoError.
"""
import ctypes

import numpy as np

try:
    from .. import _lib
    from .ray_wt_grad import _context
except ImportError:      # the package directory itself on sys.path (`import src`)
    import _lib
    from src.ray_wt_grad import _context

_vp = ctypes.c_void_p


def _vec(a, dtype, n, name):
    a = np.ascontiguousarray(a, dtype=dtype).ravel()
    if a.size < n:      # assumed-shape dummies: the Fortran reads elements 1 .. n_vox
        raise ValueError("%s has %d elements, n_vox = %d" % (name, a.size, n))
    return a


def _common(n_vox, floor_x, floor_z, alpha_x, alpha_z, ndim_x, ndim_z):
    n = int(n_vox)
    if n < 0 or int(ndim_x) <= 0 or int(ndim_z) <= 0:
        raise ValueError("n_vox >= 0 and ndim_x, ndim_z > 0 required")
    return (n, _vec(floor_x, np.int32, n, "floor_x"), _vec(floor_z, np.int32, n, "floor_z"), _vec(alpha_x, np.float32, n, "alpha_x"),
            _vec(alpha_z, np.float32, n, "alpha_z"), int(ndim_x), int(ndim_z))


def bilinear_sparse(n_vox, floor_x, floor_z, alpha_x, alpha_z, ndim_x, ndim_z):
    """src/vox_wt_grad.f90:58-112.  Returns (dat_inds, det_inds, wts, n_inds); the arrays have length 4*n_vox and are -999 beyond n_inds,
    as the Fortran leaves them; det_inds is x-fastest: fx + ndim_x * fz (:83)."""
    n, fx, fz, ax, az, ndx, ndz = _common(n_vox, floor_x, floor_z, alpha_x, alpha_z, ndim_x, ndim_z)
    dat, det, wts = np.empty(4 * n, np.int32), np.empty(4 * n, np.int32), np.empty(4 * n, np.float32)
    k = ctypes.c_int32(0)
    c = _context()
    c.check(c.lib.tomo_bilinear_sparse(c.handle, n, fx.ctypes.data_as(_vp), fz.ctypes.data_as(_vp), ax.ctypes.data_as(_vp), az.ctypes.data_as(_vp), ndx, ndz,
                                       dat.ctypes.data_as(_vp), det.ctypes.data_as(_vp), wts.ctypes.data_as(_vp), ctypes.byref(k)))
    return dat, det, wts, int(k.value)


def bilinear_vox_interp(n_vox, floor_x, floor_z, alpha_x, alpha_z, rec, ndim_x, ndim_z, der_points):
    """src/vox_wt_grad.f90:1-55.  Returns (det_img (ndim_z, ndim_x), grad_det_img (6, ndim_z, ndim_x)) float32, Fortran order, gradient rows
    tx, ty, tz, phi, alpha, beta (the order of der_points' first axis, utilities/voxel_utilities.py:38-46)."""
    n, fx, fz, ax, az, ndx, ndz = _common(n_vox, floor_x, floor_z, alpha_x, alpha_z, ndim_x, ndim_z)
    rc = _vec(rec, np.float32, n, "rec")
    der = np.asfortranarray(der_points, dtype=np.float32)
    if der.ndim != 3 or der.shape[0] != 6 or der.shape[1] != 3 or der.shape[2] < n:
        raise ValueError("der_points must have shape (6, 3, n_vox)")
    if der.shape[2] != n:
        der = np.asfortranarray(der[:, :, :n])
    img = np.empty((ndz, ndx), np.float32, order="F")
    grad = np.empty((6, ndz, ndx), np.float32, order="F")
    c = _context()
    c.check(c.lib.tomo_bilinear_vox_interp(c.handle, n, fx.ctypes.data_as(_vp), fz.ctypes.data_as(_vp), ax.ctypes.data_as(_vp), az.ctypes.data_as(_vp),
                                           rc.ctypes.data_as(_vp), ndx, ndz, der.ctypes.data_as(_vp), img.ctypes.data_as(_vp), grad.ctypes.data_as(_vp)))
    return img, grad
