#!/usr/bin/env python3
"""One-off measurement: the reference's projection_matrix at BASELINE config 1 (128^3 x 64 angles: 5.6e8 stored entries, 280 s and 35 GB peak RSS in the
reference) assembled on the device by RayOperator.tocsr() (csrc/tomo_csr.hip), and a product check against the matrix-free kernels."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tomography_alignment_amd.utilities.geometry import Geometry  # noqa: E402
from tomography_alignment_amd.utilities.projection_operators import ProjectionMatrix  # noqa: E402

N, n_proj = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 64)
geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
A = ProjectionMatrix(geo).projection_matrix()
A.dot(np.ones(N ** 3, np.float32))            # context, kernels loaded
t0 = time.perf_counter()
M = A.tocsr()
t1 = time.perf_counter()
x = np.random.default_rng(0).standard_normal(N ** 3).astype(np.float32)
a, b = A.dot(x), M.dot(x)
print("N=%d n_proj=%d: CSR with %d stored entries (%.2f GB) assembled on the device and downloaded in %.2f s; matrix-free A x vs CSR A x rel-max %.1e"
      % (N, n_proj, M.nnz, (M.data.nbytes + M.indices.nbytes + M.indptr.nbytes) / 1e9, t1 - t0, np.max(np.abs(a - b)) / np.max(np.abs(b))))
