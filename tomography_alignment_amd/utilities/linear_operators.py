"""
`utilities.linear_operators` -- imported by the reference's recon/cgls.py:3 and recon/cgls_mpi.py:5 but
absent from its snapshot.  recon/cgls.py:52 shows the intended interface
(`project(x, n_proj, angles, xyz_shift)`); this module provides it on top of the matrix-free operator,
plus the matching `backproject`, so those files import.
"""
import numpy as np

from . import projection_operators


class LinearOperator(object):

    def __init__(self, geometry, precision=np.float32, backend=None):
        self.geometry = geometry
        self.precision = precision
        self._pm = projection_operators.ProjectionMatrix(geometry, precision=precision, backend=backend)
        self._key = None
        self._op = None

    def _operator(self, n_proj, angles, xyz_shift):
        angles = np.asarray(angles, np.float64).reshape(n_proj, 3)
        xyz_shift = np.asarray(xyz_shift, np.float64).reshape(n_proj, 3)
        key = (angles.tobytes(), xyz_shift.tobytes())
        if key != self._key:
            self._op = self._pm.projection_matrix(phi=angles[:, 0], alpha=angles[:, 1], beta=angles[:, 2], xyz_shift=xyz_shift)
            self._key = key
        return self._op

    def project(self, x, n_proj, angles, xyz_shift):
        """A.x reshaped (n_proj, n_det); angles rows (phi, alpha, beta) as recon/cgls.py:31-32."""
        return self._operator(n_proj, angles, xyz_shift).dot(np.asarray(x).ravel()).reshape(n_proj, -1)

    def backproject(self, y, n_proj, angles, xyz_shift):
        """A^T.y (n_vox,)."""
        return self._operator(n_proj, angles, xyz_shift).T.dot(np.asarray(y).ravel())


def project(geometry, x, n_proj, angles, xyz_shift, precision=np.float32):
    return LinearOperator(geometry, precision).project(x, n_proj, angles, xyz_shift)


def backproject(geometry, y, n_proj, angles, xyz_shift, precision=np.float32):
    return LinearOperator(geometry, precision).backproject(y, n_proj, angles, xyz_shift)
