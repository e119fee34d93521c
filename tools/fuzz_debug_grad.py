#!/usr/bin/env python3
"""Development aid: for one seed of tests/test_gpu_fuzz.py::test_gradient_kernels_agree_on_random_geometry print, per case, pose and kernel,
the worst well-conditioned ray's error against the float64 oracle and that ray's distance to the nearest cell face; variant 1 also
with its diagnostic precisions (option grad_v1_prec).  The CPU model of the same arithmetic: tools/grad_error_model.py."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
from tomography_alignment_amd import _lib  # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry  # noqa: E402
from tomography_alignment_amd.utilities.projection_operators import ProjectionMatrix  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(7000 + seed)
for k in range(6):
    shape = tuple(int(v) for v in rng.integers(16, 72, 3))
    ndet = (int(rng.integers(5, 80)), int(rng.integers(3, 140)))
    step = float(rng.choice([1.0, 1.0, 0.5, 1.3]))
    n = int(rng.integers(1, 4))
    phi = rng.uniform(0, np.pi, n)
    tilt = np.deg2rad(rng.choice([0.0, 0.5, 2.0, 6.0]))
    alpha, beta = rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n)
    xyz = rng.uniform(-4, 4, (n, 3))
    cor = np.zeros((n, 3))
    cor[:, 0] = rng.uniform(-1, 1, n)
    geo = Geometry(n, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor, step_size=step)
    og = orc.Geo(n, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor, step_size=step)
    ii, jj, kk = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
    fr, ph = rng.uniform(0.05, 0.35, 3), rng.uniform(0, 6.28, 3)
    x = (0.6 + 0.4 * np.cos(fr[0] * ii + ph[0]) * np.cos(fr[1] * jj + ph[1]) * np.cos(fr[2] * kk + ph[2])).astype(np.float32)
    n_det = ndet[0] * ndet[1]
    want_p = np.zeros((n, n_det)); want_g = np.zeros((n, 6, n_det)); dist = np.zeros((n, n_det))
    for i in range(n):
        want_p[i], want_g[i] = orc.projection_gradient(og, x, alpha[i], beta[i], phi[i], xyz[i], cor[i], precision=np.float64)
        dist[i] = orc.ray_face_distance(og, alpha[i], beta[i], phi[i], xyz[i], cor[i])
    if np.max(np.abs(want_p)) == 0:
        continue
    rng.standard_normal(want_p.shape)
    poses = _lib.poses_array(phi, alpha, beta, xyz, cor)
    P = ProjectionMatrix(geo)
    be = P.backend
    d_x = be.upload(x)
    for v, prec in ((1, 0), (1, 1), (1, 2), (1, 3), (2, 0), (3, 0)):     # variant 1 also with float64 positions (1), float64 lerps + sums (2), both (3)
        be.ctx.set_option("grad_variant", v)
        be.ctx.set_option("grad_v1_prec", prec)
        for i in range(n):
            pr, gd = be.empty(n_det), be.empty(6 * n_det)
            be.proj_grad(poses[i:i + 1], d_x, pr, gd)
            g = gd.download().reshape(6, n_det).astype(np.float64)
            gmax = [max(np.max(np.abs(want_g[i][:3])), 1e-30)] * 3 + [max(np.max(np.abs(want_g[i][3:])), 1e-30)] * 3
            err = np.max(np.abs(g - want_g[i]) / np.array(gmax)[:, None], axis=0)
            well = dist[i] >= 2e-5
            j = int(np.argmax(np.where(well, err, 0)))
            print("case %d shape %s ndet %s tilt %.1f pose %d kernel %d prec %d: worst well-conditioned ray %d err %.2e (row-group maxima %.3g / %.3g), its face distance %.2e; rays within 1e-4 of a face: %d of %d"
                  % (k, shape, ndet, np.rad2deg(tilt), i, v, prec, j, err[j], gmax[0], gmax[3], dist[i][j], int(np.sum(dist[i] < 1e-4)), n_det), flush=True)
