#!/usr/bin/env python3
"""
Joint reconstruction + rigid-body alignment -- the loop of the reference's examples/align_rigid.py:27-59:
alternate (a) SIRT with the current pose estimates and (b) a per-projection L-BFGS-B on cost_xzab / gradient_xzab
(tx, tz, alpha, beta; bounds +-3 px / +-0.02 rad).  Here (a) is the device-resident solver and (b) aligns all
projections in lock step, one fused cost/gradient launch per round (tomography_alignment_amd.alignment).

    python -m tomography_alignment_amd.examples.align_rigid data.npz --outer 5 --sirt-iters 50
"""
import argparse
import time

import numpy as np

from .. import alignment
from ..recon import sirt
from ..utilities import geometry


def run(data, n_outer=5, sirt_iters=50, bounds=((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02)), verbose=True, backend=None, align_kwargs=None):
    proj = np.asarray(data["projections"], np.float32)
    phi = np.asarray(data["phi"], np.float64)
    ground_truth = data["phantom"] if "phantom" in data else None
    n_proj = proj.shape[0]
    nx, nz = proj.shape[1], proj.shape[2]
    ny = ground_truth.shape[1] if ground_truth is not None else nx
    geom = geometry.Geometry(n_proj, np.array([nx, ny, nz]), np.ones(3), np.array([nx, nz]), np.ones(2))
    alpha_rec, beta_rec, xyz_rec = np.zeros(n_proj), np.zeros(n_proj), np.zeros((n_proj, 3))
    rec = None
    history = []
    for it in range(n_outer):
        opts = {"_backend": backend} if backend is not None else {}
        if ground_truth is not None:
            opts["ground_truth"] = ground_truth
        if rec is not None:
            opts["rec"] = rec.ravel()                                   # warm start, examples/align_rigid.py:42
        t0 = time.perf_counter()
        solver = sirt.SIRT(geom, proj.reshape(n_proj, -1), np.array([phi, alpha_rec, beta_rec]).T, xyz_rec, options=opts)
        rec, err = solver.run_main_iteration(niter=sirt_iters, positivity=True)
        t1 = time.perf_counter()
        res = alignment.align_projections(solver.be, solver.d_rec, proj.reshape(n_proj, -1), phi, letters="xzab", bounds=bounds, **(align_kwargs or {}))
        t2 = time.perf_counter()
        xyz_rec[:, 0], xyz_rec[:, 2] = res["x"][:, 0], res["x"][:, 1]
        alpha_rec, beta_rec = res["x"][:, 2].copy(), res["x"][:, 3].copy()
        entry = {"outer": it, "rmse": float(err[-1]), "residual": float(res["fun"].sum()), "launches": res["n_launch"], "evals": res["n_eval"], "driver": res.get("driver"),
                 "sirt_wall_s": round(t1 - t0, 3), "align_wall_s": round(t2 - t1, 3)}
        if "xyz" in data:
            entry["shift_err_px"] = float(np.abs(xyz_rec[:, [0, 2]] - np.asarray(data["xyz"])[:, [0, 2]]).mean())
            entry["tilt_err_deg"] = float(np.rad2deg(np.abs(np.column_stack([alpha_rec, beta_rec]) -
                                                            np.column_stack([data["alpha"], data["beta"]])).mean()))
        history.append(entry)
        if verbose:
            print(entry)
    return rec, alpha_rec, beta_rec, xyz_rec, history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("data")
    ap.add_argument("--outer", type=int, default=5)
    ap.add_argument("--sirt-iters", type=int, default=50)
    a = ap.parse_args()
    run(dict(np.load(a.data)), a.outer, a.sirt_iters)


if __name__ == "__main__":
    main()
