"""
Voxel-driven (splat) projector with the function names and return values of the reference's
utilities/voxel_utilities.py:6-108 -- `rigid_transformation`, `derivative_rigid`, `forward_sparse`,
`forward_proj_grad` -- computed by the `k_vox_splat` kernel (src/vox_wt_grad.f90 semantics).  The reference never
calls this path (utilities/projection_operators.py:54 hard-wires the ray-driven one); it is provided because
BASELINE.json names src/vox_wt_grad.f90.  Detector index here is x-fastest (fx + ndim_x*fz), unlike the ray path.
"""
import weakref

import numpy as np

try:
    from .. import _lib
    from ..backend import HipBackend
except ImportError:      # imported as top-level `utilities`
    import _lib
    from backend import HipBackend
from .rotations import rot_x, rot_y, rot_z, der_rot_x, der_rot_y, der_rot_z

_backends = weakref.WeakKeyDictionary()


def _backend(geometry, backend=None):
    if backend is not None:
        return backend
    be = _backends.get(geometry)
    if be is None:
        be = HipBackend(geometry)
        _backends[geometry] = be
    return be


def rigid_transformation(x, alpha, beta, phi, xyz):
    """x' = Ry(beta) (Rx(alpha) Rz(phi) x + xyz)      (reference :6-20)."""
    return np.dot(rot_y(beta), np.dot(rot_x(alpha), np.dot(rot_z(phi), x)) + np.asarray(xyz)[:, np.newaxis])


def derivative_rigid(x, a, b, t, s):
    """(6, 3, n) Jacobian rows tx,ty,tz,phi(t),alpha(a),beta(b)      (reference :23-48)."""
    R_b, R_a, R_t = rot_y(b), rot_x(a), rot_z(t)
    rtx = np.dot(R_t, x)
    ratx = np.dot(R_a, rtx)
    der = np.zeros((6, x.shape[0], x.shape[1]))
    for k in range(3):
        der[k] = R_b[:, k][:, np.newaxis]
    der[3] = np.dot(np.dot(R_b, R_a), np.dot(der_rot_z(t), x))
    der[4] = np.dot(R_b, np.dot(der_rot_x(a), rtx))
    der[5] = np.dot(der_rot_y(b), ratx + np.asarray(s)[:, np.newaxis])
    return der


def _pose_cor(geometry, alpha, beta, phi, xyz_shift):
    cor = np.asarray(geometry.cor_shift, np.float64).reshape(-1)[:3]     # callers set geometry.cor_shift to one 3-vector (:61)
    pose = _lib.poses_array([phi], [alpha], [beta], np.asarray(xyz_shift, np.float64).reshape(1, 3), np.zeros(3))
    return pose, cor


def forward_sparse(geometry, alpha, beta, phi, xyz_shift, backend=None):
    """-> (dat_inds, det_inds, wts): bilinear splat triplets of one projection      (reference :51-79)."""
    be = _backend(geometry, backend)
    pose, cor = _pose_cor(geometry, alpha, beta, phi, xyz_shift)
    return be.vox_triplets(pose, cor)


def forward_proj_grad(geometry, alpha, beta, phi, xyz_shift, rec, backend=None):
    """-> (det_img.ravel(), gradient.reshape(6, -1)) float32      (reference :82-108)."""
    be = _backend(geometry, backend)
    pose, cor = _pose_cor(geometry, alpha, beta, phi, xyz_shift)
    vol = rec if be.is_buffer(rec) else be.upload(np.asarray(rec, np.float32).ravel())
    img, grad = be.empty(be.n_det), be.empty(6 * be.n_det)
    be.vox_splat(pose, cor, vol, img, grad)
    return img.download(), grad.download().reshape(6, -1)
