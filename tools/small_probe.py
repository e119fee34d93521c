import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from tomography_alignment_amd import _lib
from tomography_alignment_amd.backend import HipBackend
from tomography_alignment_amd.recon import sirt
from tomography_alignment_amd.utilities.geometry import Geometry
from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN
ctx = _lib.Context(0)
for N, n_proj, tilt in ((64, 90, 0), (64, 90, 1), (128, 64, 0), (128, 64, 1), (256, 256, 0)):
    geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    be = HipBackend(geo, ctx=ctx)
    phi = np.linspace(0, np.pi, n_proj)
    rng = np.random.default_rng(0)
    a = np.deg2rad(rng.uniform(-1, 1, n_proj)) * tilt; b = np.deg2rad(rng.uniform(-1, 1, n_proj)) * tilt
    xyz = np.zeros((n_proj, 3))
    d_true = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
    d_b = be.forward(_lib.poses_array(phi, a, b, xyz, np.zeros(3)), d_true, be.empty(n_proj * N * N))
    s = sirt.SIRT(geo, d_b, np.array([phi, a, b]).T, xyz, {"_backend": be})
    s.iterate_device(niter=5)
    ctx.sync(); ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.perf_counter(); k, _ = s.iterate_device(niter=50); ctx.sync(); dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    tot = 0.0; parts = []
    for nm in ("k_fwd_tile_flat", "k_fwd_tile", "k_adj_gather_flat", "k_adj_tile", "k_adj_tile_flat", "k_fwd_live", "k_sino_zflags", "k_absmax", "k_residual_scale", "k_update", "k_pad", "k_unpad", "k_fwd_v2", "k_adj_v1"):
        n, ms = ctx.profile_get(nm)
        if n: tot += ms; parts.append("%s %.3f" % (nm, ms / k))
    print("N=%d n_proj=%d tilt=%d: %.3f ms/iteration wall, kernels %.3f ms/iteration (%s)" % (N, n_proj, tilt, 1e3 * dt / k, tot / k, ", ".join(parts)), flush=True)
