"""
Parallel-beam geometry: same constructor and attributes as the reference's
utilities/geometry.py:9-105 `Geometry`, so reference scripts can pass it around unchanged.

Difference in construction only: `vox_centers` (3 x n_vox float64 -- 25.8 GB at 1024^3) is built
lazily on first access instead of eagerly (geometry.py:85-86); the GPU kernels never need it because
a voxel centre is `vox_origin + index*pitch`.
"""
import numpy as np


def _cell_centres(extent, n):
    # geometry.py:82-84,92-93: n cells over [-extent/2, extent/2), centres offset by a hard-coded +0.5
    return np.linspace(-extent / 2, extent / 2, n, endpoint=False) + 0.5


class Geometry(object):
    """Detector and object set-up for parallel-beam geometry (reference: utilities/geometry.py:14-47)."""

    def __init__(self, n_proj, voxel_shape, voxel_pixsize, detector_shape, detector_pixsize,
                 cor_shift=None, step_size=1.0):
        self.n_proj = n_proj
        self.vox_shape = voxel_shape
        self.vox_pix = voxel_pixsize
        self.vox_size = self.vox_shape * self.vox_pix
        self.n_vox = np.prod(self.vox_shape)
        self.det_shape = detector_shape
        self.det_pix = detector_pixsize
        self.det_size = self.det_shape * self.det_pix
        self.n_det = np.prod(self.det_shape)
        self.vox_ds = np.array([1, 1, 1])
        if cor_shift is None:
            self.cor_shift = np.zeros((n_proj, 3))
        elif np.ndim(cor_shift) == 2:
            if np.shape(cor_shift) != (n_proj, 3):
                raise AssertionError("cor_shift must be (n_proj, 3)")
            self.cor_shift = cor_shift
        elif np.ndim(cor_shift) == 1:
            if np.size(cor_shift) != 3:
                raise AssertionError("cor_shift must have 3 components")
            self.cor_shift = np.tile(cor_shift, n_proj).reshape(n_proj, 3)
        else:
            print('shape or size of cor_shift not valid')
        self.step_size = step_size
        self._vox_centers = None
        self._build_grids()

    # reference name (geometry.py:77); kept so subclasses / callers that re-run it still work
    def _voxel_detector_grid(self):
        self._build_grids()

    def _build_grids(self):
        nx, ny, nz = (int(v) for v in self.vox_shape)
        sx, sy, sz = (float(v) for v in self.vox_size)
        self._axes = (_cell_centres(sx, nx), _cell_centres(sy, ny), _cell_centres(sz, nz))
        self.vox_origin = np.array([a.min() for a in self._axes])
        self._vox_centers = None

        ndx, ndz = (int(v) for v in self.det_shape)
        dsx, dsz = (float(v) for v in self.det_size)
        xd, zd = _cell_centres(dsx, ndx), _cell_centres(dsz, ndz)
        gx = np.repeat(xd, ndz)          # ray r = ix*ndz + iz  (meshgrid 'ij' + ravel, geometry.py:94)
        gz = np.tile(zd, ndx)
        n_det = ndx * ndz
        # rays run from y = -sy to y = +sy (geometry.py:95-100)
        self.source_centers = np.array([gx, np.full(n_det, -sy), gz])
        self.det_centers = np.array([gx, np.full(n_det, sy), gz])
        # voxel-driven path bookkeeping (geometry.py:103-105)
        self.det_orig = np.array([xd.min(), self._axes[1].min(), zd.min()])
        self.factor = np.array([float(nx / ndx), 1., float(nz / ndz)])

    @property
    def vox_centers(self):
        if self._vox_centers is None:
            X, Y, Z = np.meshgrid(*self._axes, indexing='ij')
            self._vox_centers = np.array([X.ravel(), Y.ravel(), Z.ravel()])
        return self._vox_centers

    @vox_centers.setter
    def vox_centers(self, value):
        self._vox_centers = value

    def _geo_parameters(self, angles=None, shifts=None):
        """angles: (n_proj,) tomo angles or (2|3, n_proj) rows (tomo, alpha[, beta]); shifts (3, n_proj).
        Sets self.angles (3, n_proj) and self.shifts (reference: geometry.py:49-75)."""
        self.angles = np.zeros((3, self.n_proj))
        if angles is None:
            self.angles[0] = np.linspace(0., np.pi, self.n_proj)
        elif np.ndim(angles) == 1:
            assert np.size(angles) == self.n_proj
            self.angles[0] = angles
        else:
            assert angles.shape[1] == self.n_proj
            self.angles[:angles.shape[0] if angles.shape[0] <= 3 else 3] = angles[:3]
        if shifts is None:
            self.shifts = np.zeros((3, self.n_proj))
        else:
            assert shifts.shape == (3, self.n_proj)
            self.shifts = shifts
