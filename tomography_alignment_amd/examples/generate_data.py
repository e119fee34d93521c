#!/usr/bin/env python3
"""
Synthetic misaligned tomography data set -- the job of the reference's examples/generate_data.py:6-29 (64^3
Shepp-Logan, 90 angles, alpha/beta jitter of +-1 deg in 0.01-deg steps, x/z jitter of +-2 px in 0.01-px steps), with
the projections computed by the GPU forward projector and, unlike the reference script, actually written to disk
(.npz with the key names of the HDF5 layout examples/align_rigid.py:11-17 reads: projections, alpha, beta, xyz, phi,
phantom).

    python -m tomography_alignment_amd.examples.generate_data --size 64 --angles 90 --out data.npz
"""
import argparse

import numpy as np

from ..utilities import generate_phantom, geometry, projection_operators


def make(size=64, n_proj=90, seed=None, ang_deg=1.0, shift_px=2.0):
    rng = np.random.RandomState(seed)
    nx = ny = nz = size
    shepp = generate_phantom.shepp3d(nx)
    geom = geometry.Geometry(n_proj, np.array([nx, ny, nz]), np.ones(3), np.array([nx, nz]), np.ones(2))
    phi = np.linspace(0.0, np.pi, n_proj)
    a100, s100 = int(round(100 * ang_deg)), int(round(100 * shift_px))
    alpha = np.deg2rad(rng.randint(-a100, a100, n_proj) / 100)          # examples/generate_data.py:17-18
    beta = np.deg2rad(rng.randint(-a100, a100, n_proj) / 100)
    xyz = np.zeros((n_proj, 3))
    xyz[:, 0] = rng.randint(-s100, s100, n_proj) / 100                  # :22-23 (motion along the beam is invisible)
    xyz[:, 2] = rng.randint(-s100, s100, n_proj) / 100
    proj_obj = projection_operators.ProjectionMatrix(geom, precision=np.float32)
    pmat = proj_obj.projection_matrix(alpha=alpha, beta=beta, phi=phi, xyz_shift=xyz)
    proj = pmat.dot(shepp.ravel()).reshape(n_proj, nx, nz)              # :29
    return dict(projections=proj, alpha=alpha, beta=beta, xyz=xyz, phi=phi, phantom=shepp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--angles", type=int, default=90)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="data.npz")
    a = ap.parse_args()
    d = make(a.size, a.angles, a.seed)
    np.savez(a.out, **d)
    print("wrote %s: projections %s, phantom %s" % (a.out, d["projections"].shape, d["phantom"].shape))


if __name__ == "__main__":
    main()
