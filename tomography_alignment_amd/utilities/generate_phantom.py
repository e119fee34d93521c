"""
3-D Shepp-Logan test volume with the call signature of the reference's
utilities/generate_phantom.py:28-46 `shepp3d(size, dtype)` (itself tomopy-derived); used as the
synthetic input of tests and bench.py.  Written slab-by-slab along x so a 1024^3 volume needs ~50 MB of
temporaries instead of several full-size float64 coordinate grids.

Ellipsoid table: the ten (A, a, b, c, x0, y0, z0, phi, theta, psi) rows of the modified Shepp-Logan
phantom (utilities/generate_phantom.py:194-209); a voxel at normalised coordinates r in [-1,1]^3 is
inside ellipsoid k when |(R_k r - m_k) / s_k|^2 <= 1 (:136-179), values add, result clipped at 0.
Pinned bit-for-bit against the reference's output in tests (golden g7).
"""
import numpy as np

SHEPP_LOGAN = np.array([
    [1., .6900, .920, .810, 0., 0., 0., 90., 90., 90.],
    [-.8, .6624, .874, .780, 0., -.0184, 0., 90., 90., 90.],
    [-.2, .1100, .310, .220, .22, 0., 0., -108., 90., 100.],
    [-.2, .1600, .410, .280, -.22, 0., 0., 108., 90., 100.],
    [.1, .2100, .250, .410, 0., .35, -.15, 90., 90., 90.],
    [.1, .0460, .046, .050, 0., .1, .25, 90., 90., 90.],
    [.1, .0460, .046, .050, 0., -.1, .25, 90., 90., 90.],
    [.1, .0460, .023, .050, -.08, -.605, 0., 90., 90., 90.],
    [.1, .0230, .023, .020, 0., -.606, 0., 90., 90., 90.],
    [.1, .0230, .046, .020, .06, -.605, 0., 90., 90., 90.]])


def _euler(phi, theta, psi):
    p, t, s = np.radians([phi, theta, psi])
    cp, sp, ct, st, cs, ss = np.cos(p), np.sin(p), np.cos(t), np.sin(t), np.cos(s), np.sin(s)
    return np.array([[cs * cp - ct * sp * ss, cs * sp + ct * cp * ss, ss * st],
                     [-ss * cp - ct * sp * cs, -ss * sp + ct * cp * cs, cs * st],
                     [st * sp, -st * cp, ct]])


def ellipsoid_phantom(size, table, dtype='float32'):
    nx, ny, nz = (size, size, size) if np.isscalar(size) else tuple(size)
    xs, ys, zs = (np.linspace(-1., 1., n) for n in (nx, ny, nz))
    Y, Z = np.meshgrid(ys, zs, indexing='ij')
    out = np.zeros((nx, ny, nz), dtype=dtype)
    rots = [_euler(*row[7:10]) for row in table]
    for i, x in enumerate(xs):
        acc = np.zeros((ny, nz), np.float64)
        first = True
        for row, R in zip(table, rots):
            A, abc, m = row[0], row[1:4], row[4:7]
            r2 = np.zeros((ny, nz))
            for k in range(3):
                # same evaluation order as a 3x3 tensordot over (x, y, z) followed by shift and scale
                q = (R[k, 0] * x + R[k, 1] * Y + R[k, 2] * Z - m[k]) / abc[k]
                r2 += q * q
            inside = r2 <= 1.
            if first:
                acc = np.where(inside, A, 0.).astype(out.dtype).astype(np.float64)
                first = False
            else:
                # the reference adds a float64 scalar into a float32 array: float64 add, float32 store
                acc = np.where(inside, (acc + A).astype(out.dtype).astype(np.float64), acc)
        out[i] = acc
    return out


def shepp3d(size=128, dtype='float32'):
    return ellipsoid_phantom(size, SHEPP_LOGAN, dtype).clip(0, np.inf)
