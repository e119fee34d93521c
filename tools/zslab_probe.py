#!/usr/bin/env python3
"""What one rank would compute per SIRT step if the 1024^3 x 1024-angle workload were split by Z SLABS instead of by angles
 -- measured on ONE GPU with the existing kernels.

For untilted poses a ray touches the volume planes of its own detector row only (csrc/kernels_tile_flat.hip.h: the flat kernels pair
detector-z with volume-z), so rank r of P can own the planes [N r / P, N (r + 1) / P) of `rec`, `V`, and the matching detector rows of
`b`, `W`: ALL angles, 1/P of the planes, no replicated volume, no volume-sized collective (tilts add a halo of N tan(tilt) / 2 planes).
One rank's step at P = 8 is then exactly a SIRT step on a 1024 x 1024 x 128 volume with a 1024 x 128 detector and 1024 angles:

    python3 tools/zslab_probe.py [--size 1024] [--angles 1024] [--parts 8] [--dense]

runs that for EVERY slab of the Shepp-Logan volume (the slabs differ: the object lives in the planes 0.155 N .. 0.845 N, so the outer
slabs are nearly empty -- a z split is load-imbalanced where the angle split is not) and prints a markdown table: per slab the step time
and its kernels; the step of the P-rank run = the SLOWEST slab.  Compare profiles/round4_per_rank_compute.md (angle split: 50.0 ms per
rank-step at P = 8, 48.4 ms in the plain sequence, before any byte moves; 337 / 8 = 42 ms would be perfect)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tomography_alignment_amd import _lib  # noqa: E402
from tomography_alignment_amd.backend import HipBackend  # noqa: E402
from tomography_alignment_amd.recon import sirt as sirt_mod  # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry  # noqa: E402
from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN  # noqa: E402

NAMES = ("k_fwd_tile_flat", "k_fwd_live", "k_adj_gather_flat", "k_adj_tile_flat", "k_fwd_tile", "k_adj_tile", "k_sino_zflags", "k_residual_scale", "k_update", "k_absmax")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--angles", type=int, default=1024)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dense", action="store_true", help="Shepp-Logan + 0.05: no zero voxel, every slab does all the work")
    a = ap.parse_args()
    N, n_proj, P = a.size, a.angles, a.parts
    ctx = _lib.Context(0)
    phi = np.linspace(0., np.pi, n_proj)
    angles = np.array([phi, 0 * phi, 0 * phi]).T
    xyz = np.zeros((n_proj, 3))
    # the whole phantom once (device), handed to the slabs through the host: x[:, :, z0:z1] is not contiguous
    geo_full = Geometry(1, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    be_full = HipBackend(geo_full, ctx=ctx)
    full = be_full.phantom(be_full.empty(N ** 3), (N, N, N), SHEPP_LOGAN).download().reshape(N, N, N)
    if a.dense:
        full += np.float32(0.05)
    del be_full
    cuts = np.linspace(0, N, P + 1).astype(int)
    rows = []
    for r in range(P):
        z0, z1 = int(cuts[r]), int(cuts[r + 1])
        nz = z1 - z0
        geo = Geometry(n_proj, np.array([N, N, nz]), np.ones(3), np.array([N, nz]), np.ones(2))
        be = HipBackend(geo, ctx=ctx)
        d_true = be.upload(np.ascontiguousarray(full[:, :, z0:z1]).ravel())
        d_b = be.forward(_lib.poses_array(phi, 0 * phi, 0 * phi, xyz, np.zeros(3)), d_true, be.empty(n_proj * N * nz))
        s = sirt_mod.SIRT(geo, d_b, angles, xyz, {"_backend": be})
        s.iterate_device(niter=1)
        ctx.sync()
        ctx.profile_reset()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        k, rms = s.iterate_device(niter=a.steps)
        ctx.sync()
        dt = (time.perf_counter() - t0) / a.steps
        ctx.profile_enable(False)
        kk = {nm: ctx.profile_get(nm)[1] / a.steps for nm in NAMES if ctx.profile_get(nm)[0]}
        rows.append((r, z0, z1, 1e3 * dt, kk, float(np.count_nonzero(full[:, :, z0:z1])) / full[:, :, z0:z1].size, float(rms[-1])))
        print("slab %d planes [%d, %d): %.1f ms per step  %s" % (r, z0, z1, 1e3 * dt, {k_: round(v, 1) for k_, v in kk.items()}), flush=True)
        del s, d_b, d_true, be
    used = sorted({k_ for row in rows for k_ in row[4]}, key=NAMES.index)
    print()
    print("| slab (planes) | non-zero voxels | step ms | " + " | ".join("`%s`" % k_ for k_ in used) + " |")
    print("|---|---|---|" + "---|" * len(used))
    for r, z0, z1, ms, kk, nnz, rms in rows:
        print("| %d [%d, %d) | %.2f | %.1f | " % (r, z0, z1, nnz, ms) + " | ".join("%.1f" % kk.get(k_, 0.0) for k_ in used) + " |")
    worst = max(row[3] for row in rows)
    mean = float(np.mean([row[3] for row in rows]))
    print()
    print("step of a %d-rank z-split run = the slowest slab: %.1f ms (mean over slabs %.1f ms; sum %.1f ms = the whole step on one GPU done slab by slab)"
          % (P, worst, mean, mean * P))
    ctx.close()


if __name__ == "__main__":
    main()
