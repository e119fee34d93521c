#!/usr/bin/env python3
"""How often does the library's n = int(|r0| / step) differ from numpy's (the reference's)?  Development aid, CPU only:
builds a small host program around csrc/tomo_raycore.h and compares it with oracle.ray_setup over random poses of a volume that
is longer in x than in y (where the last sample of a ray lies inside the object, HISTORY.md section 2)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

SRC = r'''
#include <cstdio>
#include "%s/tomography_alignment_amd/csrc/tomo_raycore.h"
int main()
{
    TomoGeomC g{};
    double org[3];
    if (scanf("%%lf %%lf %%lf %%lf %%lf %%lf %%lf %%lf", &g.det_x0, &g.det_z0, &g.src_y, &g.det_y, &org[0], &org[1], &org[2], &g.step) != 8) return 1;
    g.det_dx = g.det_dz = 1;
    for (int a = 0; a < 3; ++a) g.org[a] = org[a];
    double p[7];
    while (scanf("%%lf %%lf %%lf %%lf %%lf %%lf %%lf", &p[0], &p[1], &p[2], &p[3], &p[4], &p[5], &p[6]) == 7) {
        ProjC c;
        tomo_make_projc(g, p, c, nullptr);
        printf("%%d %%.17g\n", c.n, c.rlen);
    }
    return 0;
}
''' % ROOT


def _fma(a, b, c):
    """Correctly rounded a * b + c (exact rational arithmetic, one rounding) -- Python 3.10 has no math.fma."""
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def _dot3(a, b):
    """The DOCUMENTED rounding of a 3-term inner product (csrc/tomo_raycore.h tomo_dot3): one rounded product, then two fused
    multiply-adds in ascending k -- what BLAS dgemm does on an FMA machine, hence what np.dot gives in the reference there."""
    return _fma(a[2], b[2], _fma(a[1], b[1], a[0] * b[0]))


def emulated_rlen_n(og, alpha, beta, phi, xyz, cor, step):
    """|r_0| and n = int(|r_0| / step) of utilities/ray_voxel_utilities.py:6-12,72-88 with every inner product rounded as
    documented above and the norm as np.linalg.norm's three rounded squares added in order: host-independent float64."""
    import math
    c, s_ = math.cos, math.sin
    Rz = [[c(phi), -s_(phi), 0.0], [s_(phi), c(phi), 0.0], [0.0, 0.0, 1.0]]
    Rx = [[1.0, 0.0, 0.0], [0.0, c(alpha), -s_(alpha)], [0.0, s_(alpha), c(alpha)]]
    Ry = [[c(beta), 0.0, s_(beta)], [0.0, 1.0, 0.0], [-s_(beta), 0.0, c(beta)]]
    col = lambda m, j: [m[0][j], m[1][j], m[2][j]]
    Rzx = [[_dot3(Rz[i], col(Rx, j)) for j in range(3)] for i in range(3)]
    ends = []
    for y in (float(og.source_centers[1, 0]), float(og.det_centers[1, 0])):
        p = [float(og.det_centers[0, 0]) + cor, y, float(og.det_centers[2, 0])]
        q = [_dot3(Ry[i], p) + float(xyz[i]) for i in range(3)]
        ends.append([_dot3(Rzx[i], q) for i in range(3)])
    org = [float(v) for v in og.vox_origin]
    r = [(ends[1][a] - org[a]) - (ends[0][a] - org[a]) for a in range(3)]
    rlen = math.sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2])
    return rlen, int(rlen / step)


def numpy_dot_is_fused():
    """Does this host's np.dot round a 3-term product the fused way?  (a0 b0 rounded, then two FMAs.)"""
    a = np.array([[1.0 + 2.0 ** -30, 1.0 + 2.0 ** -29, 1.0 + 2.0 ** -28]])
    b = np.array([[1.0 + 2.0 ** -30], [1.0 - 2.0 ** -29], [-2.0 - 2.0 ** -27]])
    return float(np.dot(a, b)[0, 0]) == _dot3(list(a[0]), list(b[:, 0]))


def main(n_poses=3000, against="both"):
    """Returns the number of poses whose n differs from the emulation of the documented rounding (`against` = "emulation"),
    from numpy's own (the oracle's ray_setup, "numpy"), or the sum of both ("both")."""
    out_dir = os.path.join(ROOT, "build", "scratch")
    os.makedirs(out_dir, exist_ok=True)
    src, exe = os.path.join(out_dir, "n_check.cpp"), os.path.join(out_dir, "n_check")
    open(src, "w").write(SRC)
    subprocess.run(["g++", "-O2", "-std=c++17", src, "-o", exe, "-lm"], check=True)
    rng = np.random.default_rng(1)
    shape, ndet, step = (54, 18, 27), (34, 90), 1.0
    phi, alpha, beta = rng.uniform(0, np.pi, n_poses), rng.uniform(-0.1, 0.1, n_poses), rng.uniform(-0.1, 0.1, n_poses)
    xyz, cor = rng.uniform(-4, 4, (n_poses, 3)), rng.uniform(-1, 1, n_poses)
    og = orc.Geo(1, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), step_size=step)
    inp = "%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" % (og.det_centers[0, 0], og.det_centers[2, 0], og.source_centers[1, 0],
                                                              og.det_centers[1, 0], *og.vox_origin, step)
    inp += "".join("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" % (phi[i], alpha[i], beta[i], *xyz[i], cor[i]) for i in range(n_poses))
    got = [ln.split() for ln in subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.splitlines()]
    n_bad = 0
    if against in ("both", "emulation"):
        emu = [emulated_rlen_n(og, alpha[i], beta[i], phi[i], xyz[i], cor[i], step) for i in range(n_poses)]
        bad_e = sum(int(g[0]) != e[1] for g, e in zip(got, emu))
        same_e = sum(float(g[1]) == e[0] for g, e in zip(got, emu))
        print("vs the documented rounding (exact-FMA emulation): n differs for %d of %d poses; |r0| bit-identical for %d" % (bad_e, n_poses, same_e))
        n_bad += bad_e + (n_poses - same_e)
    if against in ("both", "numpy"):
        ref = [orc.ray_setup(og, alpha[i], beta[i], phi[i], xyz[i], np.array([cor[i], 0, 0]))[2:4] for i in range(n_poses)]
        bad_n = sum(int(g[0]) != r[0] for g, r in zip(got, ref))
        same_n = sum(float(g[1]) == r[1] for g, r in zip(got, ref))
        print("vs this host's numpy (np.dot fused here: %s): n differs for %d of %d poses; |r0| bit-identical for %d"
              % (numpy_dot_is_fused(), bad_n, n_poses, same_n))
        n_bad += bad_n
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
