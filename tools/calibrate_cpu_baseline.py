#!/usr/bin/env python3
"""
Calibration of bench.py's CPU baseline: how does the CPU oracle -- the restatement that is
timed on the GPU box, where the reference itself never travels -- compare in speed with the REFERENCE's own compiled Fortran,
on the same host, one thread, same inputs?

Authoring container only (needs /root/reference built by oracle/build_ref.sh into oracle/_ref/):

    bash -c 'ulimit -s unlimited && python tools/calibrate_cpu_baseline.py'      # the Fortran keeps (3, n_rays, n) automatics on the stack

Timed pairs (float32 Fortran, one thread  vs  oracle, one thread):
    forward_project_   (src/forward_projection.f90:1-68)    vs  oracle.forward            -- same sums (A x)
    compute_gradient_  (src/projection_gradient.f90:1-79)   vs  oracle.projection_gradient
    back_project_      (src/back_projection.f90:1-34)       vs  oracle.back_project_voxel -- the voxel-driven back-projector
and, for the exact adjoint the solvers use (recon/sirt.py:61, scipy CSC product of the assembled matrix, no Fortran source),
the oracle's scatter adjoint is timed alone.  Prints a markdown table (copied into BASELINE.md).
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def main():
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libref_mf.so")
    if not os.path.exists(lib_path):
        raise SystemExit("oracle/_ref/libref_mf.so missing: run oracle/build_ref.sh in the authoring container")
    lib = ctypes.CDLL(lib_path)
    f32 = np.float32
    F = lambda a: np.asfortranarray(a, dtype=f32)  # noqa: E731
    I = lambda v: ctypes.byref(ctypes.c_int32(int(v)))  # noqa: E731
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    orc.set_threads(1)
    rows = []
    for N, n_proj in ((64, 90), (128, 64)):
        rng = np.random.default_rng(N)
        geo = orc.Geo(n_proj, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2))
        x = orc.shepp3d(N).astype(f32)
        phi = np.linspace(0., np.pi, n_proj)
        alpha, beta = np.deg2rad(rng.uniform(-1, 1, n_proj)), np.deg2rad(rng.uniform(-1, 1, n_proj))
        xyz = np.zeros((n_proj, 3))
        xyz[:, 0], xyz[:, 2] = rng.uniform(-2, 2, n_proj), rng.uniform(-2, 2, n_proj)
        n_rays, n_vox = geo.n_det, geo.n_vox
        al, be, ph = F(alpha), F(beta), F(phi)
        xyzT, corT = F(xyz.T), F(np.zeros((3, n_proj)))
        src, det, org = F(geo.source_centers), F(geo.det_centers), F(geo.vox_origin)
        step = ctypes.byref(ctypes.c_float(1.0))
        rec = F(x.ravel())
        ax = np.zeros((n_proj, n_rays), dtype=f32, order="F")
        t0 = time.perf_counter()
        lib.forward_project_(P(al), P(be), P(ph), P(xyzT), P(corT), P(src), P(det), P(org), step, I(N), I(N), I(N), P(rec), I(n_proj), I(n_rays),
                             I(n_vox), P(ax))
        t_ref_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        mine = orc.forward(geo, x, alpha=alpha, beta=beta, phi=phi, xyz_shift=xyz)
        t_orc_f = time.perf_counter() - t0
        err_f = np.max(np.abs(mine - np.ascontiguousarray(ax))) / np.max(np.abs(mine))
        y = mine.astype(f32)
        t0 = time.perf_counter()
        orc.adjoint(geo, y, alpha=alpha, beta=beta, phi=phi, xyz_shift=xyz)
        t_orc_a = time.perf_counter() - t0
        det_img = np.asfortranarray(y.reshape(n_proj, N, N), dtype=f32)
        vc = F(geo.vox_centers)
        atx = np.zeros(n_vox, dtype=f32)
        t0 = time.perf_counter()
        lib.back_project_(P(al), P(be), P(ph), P(xyzT), P(vc), P(org), P(det_img), I(n_proj), I(n_vox), I(N), I(N), P(atx))
        t_ref_b = time.perf_counter() - t0
        t0 = time.perf_counter()
        mine_b = orc.back_project_voxel(geo, y.reshape(n_proj, N, N), alpha, beta, phi, xyz)
        t_orc_b = time.perf_counter() - t0
        err_b = np.max(np.abs(mine_b - atx)) / np.max(np.abs(atx))
        a1 = np.zeros(n_rays, dtype=f32)
        d1 = np.zeros((6, n_rays), dtype=f32, order="F")
        n_g = 4
        t0 = time.perf_counter()
        for i in range(n_g):
            lib.compute_gradient_(ctypes.byref(ctypes.c_float(alpha[i])), ctypes.byref(ctypes.c_float(beta[i])), ctypes.byref(ctypes.c_float(phi[i + 3])),
                                  P(F(xyz[i])), P(F(np.zeros(3))), P(src), P(det), P(org), step, I(N), I(N), I(N), P(rec), I(n_rays), I(n_vox), P(a1), P(d1))
        t_ref_g = (time.perf_counter() - t0) / n_g
        g1 = orc.Geo(1, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2))
        t0 = time.perf_counter()
        for i in range(n_g):
            orc.projection_gradient(g1, x, alpha[i], beta[i], phi[i + 3], xyz[i], np.zeros(3))
        t_orc_g = (time.perf_counter() - t0) / n_g
        rows.append((N, n_proj, t_ref_f, t_orc_f, err_f, t_ref_b, t_orc_b, err_b, t_ref_g, t_orc_g, t_orc_a))
    cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][:1]
    print("host: %s, 1 thread; reference = flang -O2 build of the untouched Fortran (oracle/build_ref.sh)" % (cpu[0] if cpu else "?"))
    print("| size | routine | reference (s) | oracle (s) | oracle / reference | results agree (rel-max) |")
    print("|---|---|---|---|---|---|")
    for N, n_proj, rf, of, ef, rb, ob, eb, rg, og, oa in rows:
        tag = "%d^3 x %d" % (N, n_proj)
        print("| %s | `forward_project` (A x, all angles) | %.2f | %.2f | %.2f | %.1e |" % (tag, rf, of, of / rf, ef))
        print("| %s | `back_project` (voxel-driven, all angles) | %.2f | %.2f | %.2f | %.1e |" % (tag, rb, ob, ob / rb, eb))
        print("| %s | `compute_gradient` (one projection) | %.3f | %.3f | %.2f | |" % (tag, rg, og, og / rg))
        print("| %s | exact adjoint A^T y (oracle scatter; the reference uses a scipy CSC product of the assembled matrix) | -- | %.2f | -- | |" % (tag, oa))


if __name__ == "__main__":
    main()
