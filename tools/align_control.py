#!/usr/bin/env python3
"""
Control experiment for the joint reconstruction + alignment loop (VERDICT r2 "missing" #2): does the REFERENCE's own loop
(examples/align_rigid.py:27-59 -- SIRT with the current poses, then per-projection L-BFGS-B on cost_xzab from a ZERO start with
bounds +-3 px / +-0.02 rad) reduce the tilt error, or does it stall like this package's 512^3 run of round 1 (shift error falls,
tilt error stays ~1 deg)?  Authoring container only: imports the reference's python + f2py modules (oracle/_ref, /root/reference)
exactly as tests/golden/make_golden.py does, and runs the same data through this package's loop on the CPU stand-in backend
(tests/backends.OracleBackend).  Development aid; its output is recorded in HISTORY.md.

    python tools/align_control.py [N=32] [n_proj=24] [n_outer=4] [sirt_iters=30] [ang_deg=1.0] [shift_px=2.0] [bound_px=3] [bound_rad=0.02]
"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
_argv, sys.argv = sys.argv, sys.argv[:1]
import make_golden as mg          # noqa: E402  (sets up the reference imports: utilities, recon, src)
import numpy as np                # noqa: E402
from scipy import optimize, sparse  # noqa: E402

a = _argv[1:]
N = int(a[0]) if len(a) > 0 else 32
n_proj = int(a[1]) if len(a) > 1 else 24
n_outer = int(a[2]) if len(a) > 2 else 4
sirt_iters = int(a[3]) if len(a) > 3 else 30
ang = float(a[4]) if len(a) > 4 else 1.0
shift = float(a[5]) if len(a) > 5 else 2.0
bpx = float(a[6]) if len(a) > 6 else 3.0
brad = float(a[7]) if len(a) > 7 else 0.02
bounds = ((-bpx, bpx), (-bpx, bpx), (-brad, brad), (-brad, brad))      # examples/align_rigid.py:48: +-3 px, +-0.02 rad

rng = np.random.RandomState(3)
x = mg.generate_phantom.shepp3d(N)
geom = mg.geom(n_proj, N)
phi = np.linspace(0.0, np.pi, n_proj)
alpha = np.deg2rad(rng.randint(-int(100 * ang), int(100 * ang), n_proj) / 100)
beta = np.deg2rad(rng.randint(-int(100 * ang), int(100 * ang), n_proj) / 100)
xyz = np.zeros((n_proj, 3))
xyz[:, 0] = rng.randint(-int(100 * shift), int(100 * shift), n_proj) / 100
xyz[:, 2] = rng.randint(-int(100 * shift), int(100 * shift), n_proj) / 100
P = mg.projection_operators.ProjectionMatrix(geom, precision=np.float32)
proj = sparse.csr_matrix.dot(P.projection_matrix(alpha=alpha, beta=beta, phi=phi, xyz_shift=xyz), x.ravel()).reshape(n_proj, N, N)


def errs(xyz_rec, a_rec, b_rec):
    return (float(np.abs(xyz_rec[:, [0, 2]] - xyz[:, [0, 2]]).mean()),
            float(np.rad2deg(np.abs(np.column_stack([a_rec, b_rec]) - np.column_stack([alpha, beta])).mean())))


print("data: %d^3, %d projections, +-%g deg, +-%g px; %d outer x (%d SIRT iterations + alignment from zero, bounds %s)"
      % (N, n_proj, ang, shift, n_outer, sirt_iters, bounds))
# ---- (a) the reference's loop, its own code
a_rec, b_rec, xyz_rec = np.zeros(n_proj), np.zeros(n_proj), np.zeros((n_proj, 3))
rec = np.zeros_like(x)
t0 = time.time()
for it in range(n_outer):
    s = mg.sirt.SIRT(geom, proj.reshape(n_proj, -1), np.array([phi, a_rec, b_rec]).T, xyz_rec, options={"ground_truth": x, "rec": rec.ravel()})
    rec, err = s.run_main_iteration(niter=sirt_iters, positivity=True)
    new_a, new_b, new_xyz, fun = np.zeros(n_proj), np.zeros(n_proj), np.zeros((n_proj, 3)), np.zeros(n_proj)
    for i in range(n_proj):
        g = copy.deepcopy(geom)
        g.cor_shift = geom.cor_shift[i]
        ao = mg.alignment_functions.AlignmentUtilities(proj[i], P, g)
        res = optimize.minimize(mg.alignment_functions.cost_xzab, np.zeros(4), method="L-BFGS-B", jac=mg.alignment_functions.gradient_xzab,
                                args=(ao, rec, np.array([phi[i], 0.0, 0.0]), np.zeros(3)), bounds=bounds, options={"disp": False})
        new_xyz[i, 0], new_xyz[i, 2], new_a[i], new_b[i], fun[i] = res.x[0], res.x[1], res.x[2], res.x[3], res.fun
    a_rec, b_rec, xyz_rec = new_a, new_b, new_xyz
    e = errs(xyz_rec, a_rec, b_rec)
    print("reference loop  outer %d: rmse %.4f  residual %.4g  shift err %.3f px  tilt err %.3f deg  (at a bound: %d of %d tilts)"
          % (it, err[-1], fun.sum(), e[0], e[1], int((np.abs(np.column_stack([a_rec, b_rec])) >= brad - 1e-9).sum()), 2 * n_proj), flush=True)
print("reference loop: %.0f s" % (time.time() - t0))

# ---- (b) this package's loop (examples/align_rigid.py) on the CPU stand-in backend, same data
from backends import OracleBackend                                    # noqa: E402
from tomography_alignment_amd.examples import align_rigid               # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry         # noqa: E402
geo2 = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
t0 = time.time()
_, a2, b2, xyz2, hist = align_rigid.run(dict(projections=proj, phi=phi, phantom=x, alpha=alpha, beta=beta, xyz=xyz), n_outer=n_outer,
                                        sirt_iters=sirt_iters, bounds=bounds, verbose=False, backend=OracleBackend(geo2))
for h in hist:
    print("this package    outer %d: rmse %.4f  residual %.4g  shift err %.3f px  tilt err %.3f deg" % (h["outer"], h["rmse"], h["residual"], h["shift_err_px"], h["tilt_err_deg"]))
print("this package (CPU stand-in backend): %.0f s" % (time.time() - t0))
print("mean |true tilt| %.3f deg" % float(np.rad2deg(np.abs(np.column_stack([alpha, beta])).mean())))
