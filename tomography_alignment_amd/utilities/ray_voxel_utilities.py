"""
The ray-driven set-up functions of the reference's utilities/ray_voxel_utilities.py under their own names and return
values -- `transform_points` (:6-12), `derivative_ray_points` (:15-50), `forward_sparse` (:53-110), `forward_proj_grad`
(:113-170) -- for callers that go below `ProjectionMatrix`.  The two numpy helpers are restated; `forward_sparse` and
`forward_proj_grad` do NOT build the reference's (3, n_rays, n) float64 sample tables: they hand the pose to the library
(tomo_triplets / tomo_proj_grad: one projection = 13 lattice constants, tomo_raycore.h), which emits the same triplets in the
same order (float64 weights) and the same projection + 6-row gradient.  `forward_proj_grad` computes in float32 on the
device and returns float64 arrays like the reference (values to ~1e-6; the float64-exact route is the reference's own file
on `src.ray_wt_grad`, INTEGRATION.md level 2 1/2).  The reference's unused pure-numpy fallbacks (`ray_tracing_trilinear`,
`ray_weights_der`, :173-) are not provided.

One deliberate difference: the reference shifts `geometry.source_centers` / `det_centers` IN PLACE by `cor_shift[0]`
(:72-73, :129-130) and relies on its callers passing a deep copy (utilities/projection_operators.py:101,114); here the
geometry is left untouched and the shift is part of the pose.
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
        be = HipBackend(geometry)          # raises without the library or a GPU: no CPU fallback
        _backends[geometry] = be
    return be


def transform_points(x, alpha, beta, phi, t):
    """x' = Rz(phi) Rx(alpha) (Ry(beta) x + t)      (reference :6-12)."""
    rot_pa = np.dot(rot_z(phi), rot_x(alpha))
    xp = np.dot(rot_y(beta), x) + np.asarray(t)[:, np.newaxis]
    return np.dot(rot_pa, xp)


def derivative_ray_points(source_points, ray_vector, alpha, beta, phi, xyz_shift):
    """(9, 3, n_rays): rows 0-2 the columns of Rz Rx (d/dt), 3-5 the angle derivatives of the transformed source point,
    6-8 the same operators applied to the untransformed ray vector (the part that scales with step / ray length)      (reference :15-50)."""
    R_p, R_a, R_b = rot_z(phi), rot_x(alpha), rot_y(beta)
    dR_p, dR_a, dR_b = der_rot_z(phi), der_rot_x(alpha), der_rot_y(beta)
    R_pa, R_ab = np.dot(R_p, R_a), np.dot(R_a, R_b)
    der = np.zeros((9, 3, source_points.shape[1]))
    for k in range(3):
        der[k] = R_pa[:, k][:, np.newaxis]
    Rb_st = np.dot(R_b, source_points) + np.asarray(xyz_shift)[:, np.newaxis]
    der[3] = np.dot(dR_p, np.dot(R_a, Rb_st))
    der[4] = np.dot(R_p, np.dot(dR_a, Rb_st))
    der[5] = np.dot(R_pa, np.dot(dR_b, source_points))
    der[6] = np.dot(dR_p, np.dot(R_ab, ray_vector))[:, np.newaxis]
    der[7] = np.dot(R_p, np.dot(dR_a, np.dot(R_b, ray_vector)))[:, np.newaxis]
    der[8] = np.dot(R_pa, np.dot(dR_b, ray_vector))[:, np.newaxis]
    return der


def _pose(geometry, alpha, beta, phi, xyz_shift):
    cor = np.asarray(geometry.cor_shift, np.float64).reshape(-1)[:3]     # callers set geometry.cor_shift to one 3-vector (projection_operators.py:102)
    return _lib.poses_array([phi], [alpha], [beta], np.asarray(xyz_shift, np.float64).reshape(1, 3), cor)


def forward_sparse(geometry, alpha, beta, phi, xyz_shift, backend=None):
    """-> (dat_inds int32, det_inds int32, wts float64): the trilinear triplets of one projection in the emission order of
    src/ray_wt_grad.f90:1-92 (ray, sample, corner), already trimmed to n_inds      (reference :53-110)."""
    return _backend(geometry, backend).triplets(_pose(geometry, alpha, beta, phi, xyz_shift))


def forward_proj_grad(geometry, alpha, beta, phi, xyz_shift, rec, backend=None):
    """-> (det_img [n_rays], grad_det_img [6, n_rays]) float64, rows tx, ty, tz, phi, alpha, beta      (reference :113-170)."""
    be = _backend(geometry, backend)
    vol = rec if be.is_buffer(rec) else be.upload(np.asarray(rec, np.float32).ravel())
    img, grad = be.empty(be.n_det), be.empty(6 * be.n_det)
    be.proj_grad(_pose(geometry, alpha, beta, phi, xyz_shift), vol, img, grad)
    return img.download().astype(np.float64), grad.download().reshape(6, -1).astype(np.float64)
