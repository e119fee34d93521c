#!/usr/bin/env python3
"""The joint reconstruction + alignment loop of examples/align_rigid.py at BASELINE config 5 scale (development aid):
N^3 Shepp-Logan, n_proj projections with +-ang deg / +-shift px pose errors, alternating SIRT and lock-step alignment.
    python tools/align_rigid_bench.py [N] [n_proj] [n_outer] [sirt_iters] [ang_deg] [shift_px]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tomography_alignment_amd.examples import align_rigid, generate_data      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_proj = int(sys.argv[2]) if len(sys.argv) > 2 else 720
n_outer = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sirt_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
ang = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
shift = float(sys.argv[6]) if len(sys.argv) > 6 else 5.0
t0 = time.perf_counter()
data = generate_data.make(size=N, n_proj=n_proj, seed=3, ang_deg=ang, shift_px=shift)
print("data: %d^3, %d projections, +-%g deg, +-%g px  (%.1f s to simulate)" % (N, n_proj, ang, shift, time.perf_counter() - t0), flush=True)
b = max(3.0, shift + 1.0), max(0.02, np.deg2rad(ang) + 0.015)
t0 = time.perf_counter()
rec, a, bt, xyz, hist = align_rigid.run(data, n_outer=n_outer, sirt_iters=sirt_iters, bounds=((-b[0], b[0]), (-b[0], b[0]), (-b[1], b[1]), (-b[1], b[1])),
                                        verbose=True)
print("total %.1f s for %d outer iterations (%d SIRT iterations + one alignment pass each)" % (time.perf_counter() - t0, n_outer, sirt_iters))
