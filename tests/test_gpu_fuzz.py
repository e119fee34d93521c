"""Randomised cross-check on the GPU: the LDS tile kernels (flat and general, forward and adjoint) against the ray-driven /
global-atomic kernels (themselves pinned to the oracle and the reference goldens) over odd shapes, steps, detector sizes and
poses -- including exactly degenerate ones.  A lost or double-counted sample at a tile boundary shows up as a ~1e-3 error."""
import numpy as np
import pytest

from conftest import rel_max

pytestmark = pytest.mark.gpu


def _case(rng, k):
    shape = tuple(int(v) for v in rng.integers(5, 140, 3))
    if k % 5 == 0:
        shape = (int(rng.integers(17, 70)), int(rng.integers(17, 70)), int(rng.choice([59, 60, 61, 119, 120, 121, 180])))
    ndet = (int(rng.integers(4, 150)), int(rng.integers(4, 200)))
    if k % 3 == 0:
        ndet = (shape[0], shape[2])
    step = float(rng.choice([1.0, 1.0, 0.5, 0.75, 1.3]))
    n_proj = int(rng.integers(1, 6))
    phi = rng.uniform(0, np.pi, n_proj)
    if k % 4 == 0:
        phi[0] = rng.choice([0.0, np.pi / 2, np.pi, np.pi / 4])
    tilted = k % 2 == 0
    alpha = np.deg2rad(rng.uniform(-4, 4, n_proj)) if tilted else np.zeros(n_proj)
    beta = np.deg2rad(rng.uniform(-4, 4, n_proj)) if tilted else np.zeros(n_proj)
    if tilted and n_proj > 1:
        alpha[-1] = beta[-1] = 0.0                  # mixed call: flat + general kernels in one launch sequence
    xyz = rng.uniform(-6, 6, (n_proj, 3))
    if k % 7 == 0:
        xyz[:] = np.round(xyz)                      # integer translations: coordinates land exactly on cell faces
    cor = np.zeros((n_proj, 3))
    cor[:, 0] = rng.uniform(-2, 2, n_proj)
    return shape, ndet, step, n_proj, phi, alpha, beta, xyz, cor


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_tile_kernels_agree_with_ray_driven_kernels(seed):
    from tomography_alignment_amd.utilities.geometry import Geometry
    from tomography_alignment_amd.utilities.projection_operators import ProjectionMatrix
    rng = np.random.default_rng(1000 + seed)
    worst_f = worst_a = 0.0
    for k in range(12):
        shape, ndet, step, n_proj, phi, alpha, beta, xyz, cor = _case(rng, k + seed)
        geo = Geometry(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor, step_size=step)
        x = rng.uniform(0.1, 1.0, shape).astype(np.float32)
        y = rng.standard_normal(n_proj * ndet[0] * ndet[1]).astype(np.float32)
        P = ProjectionMatrix(geo)
        ctx = P.backend.ctx
        A = P.projection_matrix(alpha=alpha, beta=beta, phi=phi, xyz_shift=xyz)
        ctx.set_option("fwd_variant", 3)
        ctx.set_option("adj_variant", 2)
        f_tile, a_tile = A.dot(x.ravel()), A.T.dot(y)
        ctx.set_option("fwd_variant", 1)
        ctx.set_option("adj_variant", 1)
        f_ray, a_ray = A.dot(x.ravel()), A.T.dot(y)
        if np.max(np.abs(f_ray)) > 0:
            e = rel_max(f_tile, f_ray)
            worst_f = max(worst_f, e)
            assert e < 5e-6, ("forward", k, shape, ndet, step, phi, alpha, beta, xyz)
        else:
            assert np.all(f_tile == 0)
        if np.max(np.abs(a_ray)) > 0:
            e = rel_max(a_tile, a_ray)
            worst_a = max(worst_a, e)
            assert e < 5e-6, ("adjoint", k, shape, ndet, step, phi, alpha, beta, xyz)
        else:
            assert np.all(a_tile == 0)
        if k % 3 == 1:
            # the pieces the pipelined multi-GPU back-projection is made of: random x-slab splits add up to the whole, and
            # accumulate=True adds to what is there (device buffers, through the backend)
            from tomography_alignment_amd import _lib
            be = P.backend
            poses = _lib.poses_array(phi, alpha, beta, xyz, cor)
            ctx.set_option("fwd_variant", 3)
            ctx.set_option("adj_variant", 2)
            d_y, d_v = be.upload(y), be.zeros(int(np.prod(shape)))
            n_xt, _ = be.xslab_info()
            cuts = sorted(set([0, n_xt] + [int(c) for c in rng.integers(0, n_xt + 1, 2)]))
            for a, b in zip(cuts[:-1], cuts[1:]):
                be.adjoint_xslab(poses, d_y, d_v, a, b)
            assert rel_max(d_v.download(), a_tile) < 2e-6 or np.max(np.abs(a_tile)) == 0, ("xslab", k, shape, cuts)
            be.adjoint(poses, d_y, d_v, accumulate=True)
            assert rel_max(d_v.download(), 2.0 * a_tile) < 2e-6 or np.max(np.abs(a_tile)) == 0, ("accumulate", k, shape)
        # adjointness of the tile pair on the same random data
        lhs = float(np.dot(f_tile.astype(np.float64), y.astype(np.float64)))
        rhs = float(np.dot(x.ravel().astype(np.float64), a_tile.astype(np.float64)))
        size = float(np.dot(np.abs(f_tile).astype(np.float64), np.abs(y).astype(np.float64)))     # the sums cancel: compare to the terms
        assert abs(lhs - rhs) <= 1e-5 * size + 1e-12
    print("worst rel-max: forward %.2e adjoint %.2e" % (worst_f, worst_a))
