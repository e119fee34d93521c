"""Elementary rotations about the coordinate axes and their angle derivatives, float64, with the function names of the
reference's utilities/rotations.py:9-48 (rot_x / rot_y / rot_z, der_rot_x / der_rot_y / der_rot_z).

Built from one rule instead of six literal matrices: a right-handed rotation by `angle` about axis k acts in the plane of
the two other axes (i, j) = (k+1, k+2) mod 3 as [[c, -s], [s, c]]; its derivative replaces (c, s) by (-s, c) and drops
the 1 on the axis.  Convention check: Rz maps e_x to (cos, sin, 0) -- the tomographic rotation."""
import numpy as np


def _rotation(axis, angle, derivative=False):
    i, j = (axis + 1) % 3, (axis + 2) % 3
    c, s = np.cos(angle), np.sin(angle)
    if derivative:
        c, s = -s, c
    m = np.zeros((3, 3))
    if not derivative:
        m[axis, axis] = 1.0
    m[i, i], m[i, j] = c, -s
    m[j, i], m[j, j] = s, c
    return m


def rot_x(angle):
    return _rotation(0, angle)


def rot_y(angle):
    return _rotation(1, angle)


def rot_z(angle):
    return _rotation(2, angle)


def der_rot_x(angle):
    return _rotation(0, angle, derivative=True)


def der_rot_y(angle):
    return _rotation(1, angle, derivative=True)


def der_rot_z(angle):
    return _rotation(2, angle, derivative=True)
