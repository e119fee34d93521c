#!/usr/bin/env python3
"""End-to-end alignment loop at BASELINE config 5 scale (development aid): N^3 Shepp-Logan, n_proj projections simulated
with +-2 deg / +-5 px pose errors, all projections aligned in lock step from the nominal poses against the TRUE volume
(alignment.align_projections: scipy L-BFGS-B per projection, one fused cost/gradient launch per round).
    python tools/align_bench.py [N] [n_proj] [max_threads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tomography_alignment_amd import _lib, alignment                      # noqa: E402
from tomography_alignment_amd.backend import HipBackend                   # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry          # noqa: E402
from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_proj = int(sys.argv[2]) if len(sys.argv) > 2 else 720
max_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(5)
phi = np.linspace(0., np.pi, n_proj, endpoint=False)
alpha, beta = np.deg2rad(rng.uniform(-2, 2, n_proj)), np.deg2rad(rng.uniform(-2, 2, n_proj))
xyz = np.zeros((n_proj, 3))
xyz[:, 0], xyz[:, 2] = rng.uniform(-5, 5, n_proj), rng.uniform(-5, 5, n_proj)
geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
be = HipBackend(geo)
vol = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
truth = _lib.poses_array(phi, alpha, beta, xyz, np.zeros(3))
b = be.forward(truth, vol, be.empty(n_proj * N * N)).download().reshape(n_proj, N * N)
be.ctx.sync()
bounds = ((-6., 6.), (-6., 6.), (-0.05, 0.05), (-0.05, 0.05))
be.ctx.profile_reset()
be.ctx.profile_enable(True)
t0 = time.perf_counter()
res = alignment.align_projections(be, vol, b, phi, letters="xzab", bounds=bounds, max_threads=max_threads)
dt = time.perf_counter() - t0
be.ctx.profile_enable(False)
n_k, ms_k = be.ctx.profile_get("k_cost_grad")
print("k_cost_grad: %d launches, %.2f s in the kernel (HIP events)" % (n_k, ms_k * 1e-3))
got = res["x"]
want = np.column_stack([xyz[:, 0], xyz[:, 2], alpha, beta])
err = np.abs(got - want)
print("N=%d n_proj=%d max_threads=%d: %.2f s (%.2f s inside evaluation calls), %d launches, %d evals (%.0f evals/s end to end, %.1f evals/projection, max %d)"
      % (N, n_proj, max_threads, dt, res["t_eval"], res["n_launch"], res["n_eval"], res["n_eval"] / dt, res["n_eval"] / n_proj, res["nfev"].max()))
bad = np.flatnonzero(np.rad2deg(err[:, 2:]).max(axis=1) > 0.05)
print("projections with tilt error > 0.05 deg:", bad.size, "nfev of those:", res["nfev"][bad][:10], "truth (deg):", np.rad2deg(want[bad][:3, 2:]).round(2).tolist())
print("recovered: shift err mean %.2e max %.2e px; tilt err mean %.2e max %.2e deg; final cost max %.3e"
      % (err[:, :2].mean(), err[:, :2].max(), np.rad2deg(err[:, 2:]).mean(), np.rad2deg(err[:, 2:]).max(), res["fun"].max()))
