#!/usr/bin/env python3
"""Development aid: where is the SIRT iterate of the benchmark's workload exactly zero?  (The flat forward skips all-zero 16 x 16 x 128
blocks; the counters of the bench's timed step show 25 % fewer sample entries than a dense volume has.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tomography_alignment_amd import _lib  # noqa: E402
from tomography_alignment_amd.backend import HipBackend  # noqa: E402
from tomography_alignment_amd.recon import sirt as sirt_mod  # noqa: E402
from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN  # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_proj = N
geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
be = HipBackend(geo)
phi = np.linspace(0, np.pi, n_proj)
z = np.zeros(n_proj)
xyz = np.zeros((n_proj, 3))
d_true = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
d_b = be.empty(n_proj * N * N)
be.forward(_lib.poses_array(phi, z, z, xyz, np.zeros(3)), d_true, d_b)
s = sirt_mod.SIRT(geo, d_b, np.array([phi, z, z]).T, xyz, {"_backend": be})
s.iterate_device(niter=2)
x = s.d_rec.download().reshape(N, N, N)
print("iterate: %.4f of the voxels are exactly zero" % np.mean(x == 0), flush=True)
nzp = np.any(x != 0, axis=(0, 1))
print("planes with a non-zero voxel: %d .. %d" % (np.flatnonzero(nzp)[0], np.flatnonzero(nzp)[-1]))
col = np.any(x != 0, axis=2)
c = (np.arange(N) - N / 2 + 0.5)
rad = np.sqrt(c[:, None] ** 2 + c[None, :] ** 2)
for lo, hi in ((0, 0.5), (0.5, 0.9), (0.9, 1.0), (1.0, 1.2), (1.2, 1.5)):
    m = (rad >= lo * N / 2) & (rad < hi * N / 2)
    print("columns at radius %.1f..%.1f of N/2: %.4f non-zero" % (lo, hi, np.mean(col[m])))
# blocks of the flat forward: x, y tiles of 16 starting at -1, z blocks of 128 starting at 0
dead = tot = 0
for bz in range((N + 127) // 128):
    zs = x[:, :, bz * 128:(bz + 1) * 128]
    a = np.any(zs != 0, axis=2)
    pad = np.zeros((N + 1 + 16, N + 1 + 16), bool)
    pad[1:N + 1, 1:N + 1] = a
    nt = (N + 1 + 15) // 16
    for tx in range(nt):
        for ty in range(nt):
            blk = pad[tx * 16:tx * 16 + 17, ty * 16:ty * 16 + 17]
            tot += 1
            dead += not blk.any()
print("flat-forward blocks: %d of %d all zero (%.3f)" % (dead, tot, dead / tot))
