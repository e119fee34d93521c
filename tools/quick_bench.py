#!/usr/bin/env python3
"""Ad-hoc kernel timing on the GPU box (development aid; the judged numbers come from bench.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tomography_alignment_amd import _lib  # noqa: E402
if os.environ.get("TOMO_AB_LIB"):              # A/B runs: another build of the library (development aid only)
    _lib.LIB_PATH = os.environ["TOMO_AB_LIB"]
from tomography_alignment_amd.backend import HipBackend  # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry  # noqa: E402


def poses(n_proj, tilt, rng, shift=False):
    phi = np.linspace(0, np.pi, n_proj)
    if shift:                           # untilted, translated: tx, tz ~ U(+-2 px) -- the flat kernels with fractional z offsets per projection
        a = np.zeros(n_proj); b = np.zeros(n_proj)
        xyz = np.zeros((n_proj, 3)); xyz[:, 0] = rng.uniform(-2, 2, n_proj); xyz[:, 2] = rng.uniform(-2, 2, n_proj)
    elif tilt:
        a = np.deg2rad(rng.uniform(-tilt, tilt, n_proj)); b = np.deg2rad(rng.uniform(-tilt, tilt, n_proj))     # tilt = half-range in degrees
        xyz = np.zeros((n_proj, 3)); xyz[:, 0] = rng.uniform(-2, 2, n_proj); xyz[:, 2] = rng.uniform(-2, 2, n_proj)
    else:
        a = np.zeros(n_proj); b = np.zeros(n_proj); xyz = np.zeros((n_proj, 3))
    return _lib.poses_array(phi, a, b, xyz, np.zeros(3))


def run(N, n_proj, what, tilt=True, reps=2, opts=None, shepp=False, nz=None, shift=False):
    rng = np.random.default_rng(0)
    nz = nz or N
    geo = Geometry(n_proj, np.array([N, N, nz]), np.ones(3), np.array([N, nz]), np.ones(2))
    be = HipBackend(geo)
    for k, v in (opts or {}).items():
        be.ctx.set_option(k, v)
    vol = be.zeros(N * N * nz); be.fill(vol, 1.0)
    if shepp == 2:                      # ones in the planes [0.155 N, 0.845 N), zero above and below: the support of the benchmark's SIRT iterate
        x = np.ones((N, N, N), np.float32)
        x[:, :, :int(0.155 * N)] = 0
        x[:, :, int(0.845 * N):] = 0
        del vol
        vol = be.upload(x.ravel())
        del x
    elif shepp:
        from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN
        be.phantom(vol, (N, N, N), SHEPP_LOGAN)
    prj = be.zeros(n_proj * N * nz); be.fill(prj, 1.0)
    P = poses(n_proj, tilt, rng, shift)
    grad = be.empty(6 * N * N) if what == "pg" else None
    fn = {"fwd": lambda: be.forward(P, vol, prj), "adj": lambda: be.adjoint(P, prj, vol),
          "bpv": lambda: be.backproject_voxel(P, prj, vol),
          "cg": lambda: be.cost_grad(P, vol, prj),
          "pg": lambda: [be.proj_grad(P[i:i + 1], vol, prj, grad) for i in range(n_proj)]}[what]
    fn(); be.sync()
    best = 1e30
    for _ in range(reps):
        be.ctx.timer_start(); fn(); ms = be.ctx.timer_stop(); best = min(best, ms)
    alg = {"fwd": 4 * N ** 3 + 4 * N * N, "adj": 8 * N ** 3 + 4 * N * N, "bpv": 8 * N ** 3 + 4 * N * N,
           "cg": 4 * N ** 3 + 4 * N * N + 28, "pg": 4 * N ** 3 + 28 * N * N}[what] * n_proj
    print("%-4s N=%4d n_proj=%4d tilt=%g shepp=%d opts=%s : %9.2f ms  alg %.1f GB/s  (%.1f%% of 8 TB/s)  [%.3f ms/angle]"
          % (what, N, n_proj, tilt, int(shepp), opts, best, alg / best / 1e6, alg / best / 1e6 / 80.0, best / n_proj), flush=True)
    del vol, prj
    be.ctx.close()


if __name__ == "__main__":
    jobs = sys.argv[1:] or ["fwd:256:256", "fwd:512:64", "fwd:1024:16", "adj:256:32", "bpv:256:64"]
    for j in jobs:
        parts = j.split(":")
        what, N, n_proj = parts[0], int(parts[1]), int(parts[2])
        opts = {}
        for kv in parts[3:]:
            k, v = kv.split("=")
            if k in ("tilt", "shepp", "nz", "shift"):
                continue
            opts[k] = int(v)
        tilt = 1.0
        for kv in parts[3:]:
            if kv.startswith("tilt="):
                tilt = float(kv[5:])
        run(N, n_proj, what, tilt=tilt, opts=opts, shepp=max([int(kv[6:]) for kv in parts[3:] if kv.startswith('shepp=')] or [0]),
            nz=max([int(kv[3:]) for kv in parts[3:] if kv.startswith('nz=')] or [0]) or None, shift=any(kv == 'shift=1' for kv in parts[3:]))
