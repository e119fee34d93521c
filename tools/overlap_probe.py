#!/usr/bin/env python3
"""Can the flat forward (bound by the memory side's float atomics, CUs 0.6 busy) and the gather back-projection (bound by the CUs' LDS / VALU,
no atomics) run SIDE BY SIDE faster than one after the other?  (round 5 probe; development aid)

Two contexts = two HIP streams on one GPU; the same 1024^3 x n-angle workload: `A x` on one, `A^T y` on the other, (a) one after the
other, (b) both in flight.  If (b) < (a), a SIRT iteration whose next forward projection runs slab by slab behind the current
back-projection (the sharded solver's pipeline, two compute streams) gains on ONE GPU as well.

    python3 tools/overlap_probe.py [--size 1024] [--angles 256]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tomography_alignment_amd import _lib  # noqa: E402
from tomography_alignment_amd.backend import HipBackend  # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry  # noqa: E402
from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--angles", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--masks", action="store_true")
    ap.add_argument("--dense", action="store_true", help="Shepp-Logan + 0.05: no zero voxel, every partial sum goes out as an atomic")
    a = ap.parse_args()
    N, n = a.size, a.angles
    geo = Geometry(n, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    phi = np.linspace(0., np.pi, 1024)[:n]
    poses = _lib.poses_array(phi, 0 * phi, 0 * phi, np.zeros((n, 3)), np.zeros(3))
    ca, cb = _lib.Context(0), _lib.Context(0)
    A, B = HipBackend(geo, ctx=ca), HipBackend(geo, ctx=cb)
    vol_a = A.phantom(A.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
    if a.dense:
        one = A.empty(N ** 3)
        A.fill(one, 0.05)
        A.axpy(vol_a, one, 1.0)
        del one
    proj_a = A.empty(n * N * N)
    proj_b = B.empty(n * N * N)
    vol_b = B.empty(N ** 3)
    A.forward(poses, vol_a, proj_a)
    ca.sync()
    proj_b.upload(proj_a.download())          # a real sinogram for the back-projection (other context: through the host)
    B.adjoint(poses, proj_b, vol_b)
    cb.sync()

    def run(concurrent):
        best = 1e9
        for _ in range(a.reps):
            ca.sync()
            cb.sync()
            t0 = time.perf_counter()
            A.forward(poses, vol_a, proj_a)
            if not concurrent:
                ca.sync()
            t1 = time.perf_counter()
            B.adjoint(poses, proj_b, vol_b)
            ca.sync()
            cb.sync()
            best = min(best, time.perf_counter() - t0)
            launch = t1 - t0
        return best, launch

    ca.sync()
    t0 = time.perf_counter()
    A.forward(poses, vol_a, proj_a)
    ca.sync()
    tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    B.adjoint(poses, proj_b, vol_b)
    cb.sync()
    tb = time.perf_counter() - t0
    seq, _ = run(False)
    con, launch = run(True)
    print("N=%d angles=%d: forward alone %.1f ms, back-projection alone %.1f ms, one after the other %.1f ms, both in flight %.1f ms (%.2fx); "
          "host time to issue the forward %.2f ms" % (N, n, 1e3 * tf, 1e3 * tb, 1e3 * seq, 1e3 * con, seq / con, 1e3 * launch), flush=True)
    if not a.masks:
        return

    # ---- with CU masks (tomo_ctx_set_cu_mask): how each kernel scales with the CUs it gets, and both side by side on disjoint sets
    def timed(be, ctx, fn):
        fn()
        ctx.sync()
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter()
            fn()
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best

    n_cu = 256
    # (a mask that keeps some CUs of EVERY group of 8 -- `(c % 8) < m` -- had no effect at all in round 5's run: only masks of whole leading
    #  ranges restrict the stream on this runtime)
    layouts = {"first": lambda k: range(k)}
    for name, pick in layouts.items():
        for k in (256, 224, 192, 160, 128):
            cus = list(pick(k))
            if not cus:
                continue
            ca.set_cu_mask(cus)
            cb.set_cu_mask(cus)
            print("mask %-8s %3d CUs: forward %.1f ms, back-projection %.1f ms" % (name, len(cus), timed(A, ca, lambda: A.forward(poses, vol_a, proj_a)),
                                                                                 timed(B, cb, lambda: B.adjoint(poses, proj_b, vol_b))), flush=True)
    for name, pick in layouts.items():
        for k in (224, 192, 160, 128):
            f_cus = list(pick(k))
            b_cus = [c for c in range(n_cu) if c not in set(f_cus)]
            ca.set_cu_mask(f_cus)
            cb.set_cu_mask(b_cus)
            tfm = timed(A, ca, lambda: A.forward(poses, vol_a, proj_a))
            tbm = timed(B, cb, lambda: B.adjoint(poses, proj_b, vol_b))
            con, _ = run(True)
            print("disjoint %-8s forward on %3d CUs (%.1f ms alone), back-projection on %3d (%.1f ms alone): both in flight %.1f ms  [unmasked one after the other %.1f]"
                  % (name, len(f_cus), tfm, len(b_cus), tbm, 1e3 * con, 1e3 * seq), flush=True)
    ca.set_cu_mask(None)
    cb.set_cu_mask(None)


if __name__ == "__main__":
    main()
