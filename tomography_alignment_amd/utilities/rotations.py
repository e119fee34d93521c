"""3x3 rotation matrices and their angle derivatives (API of the reference's
utilities/rotations.py:9-48; float64).  Rz = tomographic axis, Rx / Ry = tilt axes."""
import numpy as np


def _cs(angle):
    return np.cos(angle), np.sin(angle)


def rot_x(angle):
    c, s = _cs(angle)
    return np.array([[1., 0., 0.], [0., c, -s], [0., s, c]])


def rot_y(angle):
    c, s = _cs(angle)
    return np.array([[c, 0., s], [0., 1., 0.], [-s, 0., c]])


def rot_z(angle):
    c, s = _cs(angle)
    return np.array([[c, -s, 0.], [s, c, 0.], [0., 0., 1.]])


def der_rot_x(angle):
    c, s = _cs(angle)
    return np.array([[0., 0., 0.], [0., -s, -c], [0., c, -s]])


def der_rot_y(angle):
    c, s = _cs(angle)
    return np.array([[-s, 0., c], [0., 0., 0.], [-c, 0., -s]])


def der_rot_z(angle):
    c, s = _cs(angle)
    return np.array([[-s, -c, 0.], [c, -s, 0.], [0., 0., 0.]])
