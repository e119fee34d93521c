import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tomography_alignment_amd.utilities.geometry import Geometry
from tomography_alignment_amd.utilities.projection_operators import ProjectionMatrix
from oracle import oracle as orc
N = 32
y = np.load(os.path.join(ROOT, "tests/golden/g2_fwd_adj.npz"))["y"]
for phis in ([0.0], [np.pi], [np.pi / 2], [0.6283185307179586], np.linspace(0, np.pi, 6)):
    phis = np.asarray(phis); n = phis.size
    geo = Geometry(n, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2))
    og = orc.Geo(n, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2))
    for v in (1, 2):
        P = ProjectionMatrix(geo); P.backend.ctx.set_option("adj_variant", v)
        A = P.projection_matrix(phi=phis)
        mine = A.T.dot(y[:n].ravel()).reshape(N, N, N)
        want = orc.adjoint(og, y[:n], phi=phis).reshape(N, N, N)
        d = np.abs(mine - want); i = np.unravel_index(np.argmax(d), d.shape)
        print("phi", np.round(phis, 3), "variant", v, "relmax", d.max() / np.abs(want).max(), "at", i, "mine", mine[i], "want", want[i])
        if d.max() / np.abs(want).max() > 1e-4:
            bad = np.argwhere(d > 1e-4 * np.abs(want).max())
            print("   n_bad", len(bad), "x range", bad[:, 0].min(), bad[:, 0].max(), "y range", bad[:, 1].min(), bad[:, 1].max(), "z range", bad[:, 2].min(), bad[:, 2].max())
        x = np.random.default_rng(0).uniform(0, 1, (N, N, N)).astype(np.float32)
        f = A.dot(x.ravel()); fw = orc.forward(og, x, phi=phis).ravel()
        print("      forward(random x) relmax", np.abs(f - fw).max() / np.abs(fw).max())
