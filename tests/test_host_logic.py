"""CPU-only tests of the host side: the C-ABI library loads and exports every declared symbol, the package
fails loudly without a GPU, and the host logic (geometry, pose packing, scipy protocol, solver control flow,
alignment API) reproduces the reference's golden vectors when the device backend is replaced by the
oracle-backed stand-in of tests/backends.py."""
import os
import re
import ctypes
import sys

import numpy as np
import pytest
from scipy import sparse, optimize

from conftest import ROOT, golden, rel_max
from backends import OracleBackend

from tomography_alignment_amd import _lib
from tomography_alignment_amd.utilities.geometry import Geometry
from tomography_alignment_amd.utilities import projection_operators, alignment_functions, rotations
from tomography_alignment_amd.recon import sirt, cgls


def geom(n_proj, N, cor_shift=None, step=1.0, ndet=None):
    ndet = N if ndet is None else ndet
    return Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([ndet, ndet]), np.ones(2), cor_shift=cor_shift, step_size=step)


@pytest.fixture(scope="session")
def built_lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_c_abi_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "tomo.h")).read()
    declared = set(re.findall(r"^TOMO_API\s+[\w\s\*]*?\b(tomo_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 40
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert built_lib.tomo_abi_version() == 1


def test_no_gpu_means_loud_failure(built_lib):
    n = ctypes.c_int(0)
    rc = built_lib.tomo_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.TomoError):
        _lib.Context()
    with pytest.raises(_lib.TomoError):
        projection_operators.ProjectionMatrix(geom(2, 8)).projection_matrix()   # no silent CPU fallback


def test_geometry_check_guards_without_a_gpu(built_lib):
    """tomo_check_geometry (the validation inside tomo_set_geometry) runs on the host: bad shapes, the 2^31-voxel limit, and
    the wide-row flag that keeps the 24-bit-multiply kernels away from slabs whose x-row pitch is >= 2^23 bytes."""
    def check(shape, ndet=(8, 8), step=1.0):
        g = _lib.TomoGeom()
        g.nx, g.ny, g.nz = shape
        g.ndx, g.ndz = ndet
        g.src_y, g.det_y, g.step = -10.0, 10.0, step
        g.det_dx = g.det_dz = 1.0
        flags = ctypes.c_int(-1)
        return built_lib.tomo_check_geometry(ctypes.byref(g), ctypes.byref(flags)), flags.value
    assert check((64, 64, 64)) == (0, 0)
    assert check((1024, 1024, 1024)) == (0, 0)                 # row pitch 1028 * 1028 * 4 = 4.2 MB < 2^23
    assert check((16, 1440, 1440)) == (0, 0)                   # 1444 * 1444 * 4 = 8 340 544 < 2^23 = 8 388 608
    assert check((16, 1445, 1445)) == (0, _lib.GEOM_WIDE_ROWS)  # 1449 * 1449 * 4 = 8 398 404 >= 2^23
    assert check((16, 4096, 4096)) == (0, _lib.GEOM_WIDE_ROWS)
    g = _lib.TomoGeom()
    g.nx = g.ny = g.nz = g.ndx = g.ndz = 32
    g.src_y, g.det_y, g.step, g.det_dx, g.det_dz = -32.0, 32.0, 1.0, 1.0, 1.5      # detector-z pitch > 1 voxel: plain kernels too
    fl = ctypes.c_int(0)
    assert built_lib.tomo_check_geometry(ctypes.byref(g), ctypes.byref(fl)) == 0 and fl.value == _lib.GEOM_WIDE_ROWS
    # sample step > 1 voxel: in-block cells leave the +-17 range the biased unsigned offsets of the SGPR-base kernels cover
    assert check((64, 64, 64), step=1.0) == (0, 0) and check((64, 64, 64), step=0.5) == (0, 0)
    assert check((64, 64, 64), step=1.3) == (0, _lib.GEOM_WIDE_ROWS) and check((64, 64, 64), step=6.0) == (0, _lib.GEOM_WIDE_ROWS)
    assert check((2048, 2048, 2048))[0] == -5                  # TOMO_ERR_UNSUPPORTED: padded volume >= 2^31 voxels
    assert check((0, 8, 8))[0] == -2 and check((8, 8, 8), step=0.0)[0] == -2
    assert check((2048, 2048, 2048))[0] == -5 and b"2^31" in built_lib.tomo_last_error(None)


def test_geometry_conventions():
    g = geom(3, 8)
    assert np.allclose(g.vox_origin, [-3.5, -3.5, -3.5])                 # geometry.py:82-87
    assert g.source_centers.shape == (3, 64) and np.all(g.source_centers[1] == -8) and np.all(g.det_centers[1] == 8)
    assert np.array_equal(g.source_centers[2, :8], g.vox_origin[2] + np.arange(8))      # z fastest: r = ix*ndz + iz
    assert g.vox_centers.shape == (3, 512) and np.array_equal(g.vox_centers[:, 1], [-3.5, -3.5, -2.5])
    assert g.cor_shift.shape == (3, 3)
    g2 = geom(4, 8, cor_shift=np.array([0.5, 0., 0.]))
    assert g2.cor_shift.shape == (4, 3) and np.all(g2.cor_shift[:, 0] == 0.5)
    s = _lib.geom_struct(geom(2, 16, step=0.5, ndet=12))
    assert (s.nx, s.ndx, s.ndz) == (16, 12, 12) and s.step == 0.5 and s.det_dx == 1.0 and s.src_y == -16 and s.det_y == 16
    assert np.isclose(s.det_x0, -5.5) and np.isclose(s.vox_origin[0], -7.5)


def test_rotations_and_pose_packing():
    a = 0.3
    for R, dR in ((rotations.rot_x, rotations.der_rot_x), (rotations.rot_y, rotations.der_rot_y), (rotations.rot_z, rotations.der_rot_z)):
        assert np.allclose(R(a) @ R(a).T, np.eye(3))
        assert np.allclose((R(a + 1e-6) - R(a - 1e-6)) / 2e-6, dR(a), atol=1e-8)
    p = _lib.poses_array([0.1, 0.2], [1, 2], [3, 4], np.arange(6).reshape(2, 3), np.array([[9., 0, 0], [8., 0, 0]]))
    assert p.shape == (2, 7) and np.array_equal(p[1], [0.2, 2, 4, 3, 4, 5, 8])
    p = _lib.poses_array([0.1], [0.], [0.], np.zeros(3), np.array([7., 1., 2.]))
    assert p[0, 6] == 7.


def test_scipy_unbound_protocol_and_mask_with_injected_backend(shepp32):
    g = golden("g2_fwd_adj")
    geo = geom(6, 32)
    P = projection_operators.ProjectionMatrix(geo, backend=OracleBackend(geo))
    A = P.projection_matrix(alpha=g["alpha"], beta=g["beta"], phi=g["phi"], xyz_shift=g["xyz"])
    assert A.shape == (6 * 1024, 32 ** 3) and P.n_proj == 6 and P.angles.shape == (6, 3)
    Ax = sparse.csr_matrix.dot(A, shepp32.ravel())                                     # recon/sirt.py:59
    ATy = sparse.csc_matrix.dot(sparse.csr_matrix.transpose(A), g["y"].ravel())        # recon/sirt.py:61
    assert rel_max(Ax, g["Ax"]) < 2e-6 and rel_max(ATy, g["ATy"]) < 2e-6
    assert rel_max(A.T.T.dot(shepp32.ravel()), g["Ax"]) < 2e-6
    with pytest.raises(ValueError):
        A.dot(np.zeros(5))
    # G1 case c: voxel mask + float64 precision + step 0.5 + 12x12 detector
    g1 = golden("g1_operator")
    ref = sparse.csr_matrix((g1["c_data"], g1["c_indices"], g1["c_indptr"]), shape=tuple(g1["c_shape"]))
    geo = geom(2, 16, step=0.5, ndet=12)
    A = projection_operators.ProjectionMatrix(geo, precision=np.float64, backend=OracleBackend(geo)).projection_matrix(
        alpha=g1["c_alpha"], beta=g1["c_beta"], phi=g1["c_phi"], xyz_shift=g1["c_xyz"], voxel_mask=g1["c_mask"])
    x = np.random.default_rng(0).standard_normal(16 ** 3)
    y = np.random.default_rng(1).standard_normal(2 * 144)
    assert A.dot(x).dtype == np.float64
    assert rel_max(A.dot(x), ref.dot(x)) < 2e-6 and rel_max(A.T.dot(y), ref.T.dot(y)) < 2e-6


@pytest.mark.parametrize("tag,positivity,use_gt", [("plain", False, False), ("pos_gt", True, True)])
def test_sirt_control_flow_vs_reference_golden(shepp32, tag, positivity, use_gt):
    g = golden("g5_sirt")
    geo = geom(16, 32)
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    opts = {"_backend": OracleBackend(geo)}
    if use_gt:
        opts["ground_truth"] = shepp32.copy()
    s = sirt.SIRT(geo, g["b"].copy(), angles, g["xyz"], options=opts)
    assert rel_max(s.W, g["W"]) < 2e-6 and rel_max(s.V, g["V"]) < 2e-6          # recon/sirt.py:33-40
    rec, err = s.run_main_iteration(niter=10, positivity=positivity)
    assert rec.shape == (32, 32, 32) and err.shape == (10,)
    assert rel_max(rec, g["rec_" + tag]) < 3e-5
    assert np.allclose(err, g["err_" + tag], rtol=3e-5)


def test_sirt_semi_convergence_stop_and_zero_guard():
    geo = geom(4, 8)
    be = OracleBackend(geo)
    angles = np.zeros((4, 3))
    angles[:, 0] = np.linspace(0, np.pi, 4)
    b = np.zeros((4, 64), np.float32)
    b[:, 10] = 1.0
    s = sirt.SIRT(geo, b, angles, np.zeros((4, 3)), options={"_backend": be, "ground_truth": np.ones(512, np.float32)})
    assert np.all(np.isfinite(s.V)) and np.all(np.isfinite(s.W))                    # 0 -> inf -> 0 guard
    rec, err = s.run_main_iteration(niter=30)
    assert len(err) <= 30 and (len(err) == 30 or err[-1] > err[-2])                 # stops when rms rises (k > 0)


def test_cgls_control_flow_vs_restated_reference(shepp32):
    from oracle import oracle as orc
    g = golden("g5_sirt")
    geo = geom(16, 32)
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    og = orc.Geo(16, np.array([32] * 3), np.ones(3), np.array([32, 32]), np.ones(2))
    kw = dict(alpha=g["alpha"], beta=g["beta"], phi=g["phi"], xyz_shift=g["xyz"])
    want, want_err = orc.cgls(lambda x: orc.forward(og, x, **kw).astype(np.float32).ravel(),
                              lambda y: orc.adjoint(og, y, **kw).astype(np.float32), 32 ** 3, g["b"], 6)
    c = cgls.CGLS(geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(geo)})
    rec, err = c.run_main_iteration(niter=6)
    assert rec.shape == (32 ** 3,)
    assert rel_max(rec, want) < 1e-4 and np.allclose(err, want_err, rtol=1e-4)
    assert err[-1] < err[0]


def _g11_swap_case(g, tag, backend_of):
    """The package's CGLS class through golden G11's operator swap (what make_golden.py::g11 does to the reference's class)."""
    geo = geom(6, 16)
    angles = np.array([g[tag + "_phi"], g[tag + "_alpha"], g[tag + "_beta"]]).T
    c = cgls.CGLS(geo, g[tag + "_b"].copy(), angles, g[tag + "_xyz"], options=backend_of(geo))
    rec, err1 = c.run_main_iteration(niter=int(g[tag + "_first"]))
    c.xyz_shift = g[tag + "_xyz2"]
    c.proj_mat = c.f_proj_obj.projection_matrix(phi=angles[:, 0], alpha=angles[:, 1], beta=angles[:, 2], xyz_shift=g[tag + "_xyz2"])
    rec, err = c.run_main_iteration(niter=12)
    return err1, rec, err


def test_cgls_class_vs_reference_golden_g11(shepp32, capsys):
    """The package's recon/cgls.py::CGLS (CPU stand-in backend) against the reference's own class run on its own CSR (golden G11):
    plain runs, the re-initialisation that continues (c) and the one that quits at k = 1 (d) -- control flow, printed lines, and
    the `_r -= alpha * r` after a re-initialisation (recon/cgls.py:60-70)."""
    g5, g = golden("g5_sirt"), golden("g11_cgls")
    geo = geom(16, 32)
    angles = np.array([g5["phi"], g5["alpha"], g5["beta"]]).T
    for tag, gt in (("a", None), ("b", shepp32)):
        opts = {"_backend": OracleBackend(geo)}
        if gt is not None:
            opts["ground_truth"] = gt.copy()
        rec, err = cgls.CGLS(geo, g5["b"].copy(), angles, g5["xyz"], options=opts).run_main_iteration(niter=10)
        assert rel_max(rec, g["rec_" + tag]) < 2e-5 and np.allclose(err, g["err_" + tag], rtol=2e-5)
    for tag in ("c", "d"):
        capsys.readouterr()
        err1, rec, err = _g11_swap_case(g, tag, lambda geo_: {"_backend": OracleBackend(geo_)})
        said = capsys.readouterr().out
        assert np.allclose(err1, g["err1_" + tag], rtol=2e-5)
        assert len(err) == len(g["err_" + tag]) and said.count("reinitializing") == int(g[tag + "_reinit_lines"]) and int("quitting" in said) == int(g[tag + "_quit"])
        assert rel_max(rec, g["rec_" + tag]) < 5e-5 and np.allclose(err, g["err_" + tag], rtol=5e-5)


def _alignment_setup(shepp32):
    g = golden("g6_alignment")
    geo = geom(1, 32)
    import copy
    this_geo = copy.copy(geo)
    this_geo.cor_shift = geo.cor_shift[0]
    P = projection_operators.ProjectionMatrix(geo, backend=OracleBackend(geo))
    ao = alignment_functions.AlignmentUtilities(g["b"].reshape(32, 32), P, this_geo)
    return g, P, ao, (ao, shepp32, np.array([float(g["phi0"]), 0., 0.]), np.zeros(3))


def test_alignment_cost_gradient_pairs_vs_reference_golden(shepp32):
    g, P, ao, args = _alignment_setup(shepp32)
    af = alignment_functions
    for tag in ("zero", "gen"):
        p = g["p_" + tag]
        assert np.isclose(af.cost_xzab(p, *args), g["cost_xzab_" + tag], rtol=1e-5, atol=1e-9)
        assert rel_max(af.gradient_xzab(p, *args), g["grad_xzab_" + tag]) < 1e-5
        p5 = np.array([p[0], p[1], 0.003, p[2], p[3]])
        assert np.isclose(af.cost_xzpab(p5, *args), g["cost_xzpab_" + tag], rtol=1e-5, atol=1e-9)
        assert rel_max(af.gradient_xzpab(p5, *args), g["grad_xzpab_" + tag]) < 1e-5
    pg = g["p_gen"]
    sc = np.array([1.0, 2.0, 50.0, 25.0])
    assert rel_max(af.gradient_xzab(pg, *args, scale_factor=sc), g["grad_xzab_scaled"]) < 1e-5
    assert rel_max(af.gradient_xzab(pg, *args, return_vector=True), g["grad_xzab_vec"]) < 1e-5
    assert rel_max(af.cost_xzab(pg, *args, return_vector=True), g["cost_xzab_vec"]) < 1e-5
    for nm, p in (("xz", [0.4, -0.7]), ("x", [0.4]), ("z", [-0.7]), ("ab", [0.004, -0.006]), ("a", [0.004]), ("b", [-0.006]),
                  ("xzb", [0.4, -0.7, -0.006])):
        p = np.array(p)
        assert np.isclose(getattr(af, "cost_" + nm)(p, *args), g["cost_" + nm], rtol=1e-5), nm
        assert rel_max(getattr(af, "gradient_" + nm)(p, *args), g["grad_" + nm]) < 1e-5, nm
    # the reference's finite-difference checkers agree with the analytic gradient to FD accuracy
    fd = af.gradient_xz_fd(np.array([0.4, -0.7]), *args)
    assert rel_max(fd, g["grad_xz"]) < 5e-2


def test_fused_evaluation_is_memoised_only_while_pinned(shepp32):
    g, P, ao, args = _alignment_setup(shepp32)
    be = P.backend
    p = g["p_gen"]
    # unpinned: the volume is re-read on every call, like the reference (utilities/projection_operators.py:112-122)
    alignment_functions.cost_xzab(p, *args)
    alignment_functions.gradient_xzab(p, *args)
    assert be.calls["cost_grad"] == 2
    # pinned ("I will not modify rec"): fun(x) then jac(x) = one fused launch
    P.pin_volume(args[1])
    c0 = alignment_functions.cost_xzab(p, *args)
    alignment_functions.gradient_xzab(p, *args)
    assert be.calls["cost_grad"] == 3
    alignment_functions.gradient_xzab(p + 1e-3, *args)
    assert be.calls["cost_grad"] == 4
    P.invalidate_volume()                      # "I did modify it": evaluated afresh, still pinned afterwards
    alignment_functions.cost_xzab(p + 1e-3, *args)
    assert be.calls["cost_grad"] == 5 and P.volume_is_pinned(args[1])
    P.unpin_volume()
    assert np.isclose(alignment_functions.cost_xzab(p, *args), c0, rtol=1e-12) and be.calls["cost_grad"] == 6


def test_another_volume_between_calls_on_a_pinned_one():
    """pin(A); evaluate B (unpinned); evaluate A again -- A's result must come from A's data, the library
    must be told to re-stage (reuse_staged_volume = 0) because B went through its padded copy in between, and the memo
    must not hand out a value computed before."""
    N = 16
    geo = geom(1, N)
    be = OracleBackend(geo)

    class Ctx(object):                                   # records what pinned_call tells the library
        def __init__(self):
            self.log = []

        def set_option(self, name, value):
            self.log.append((name, value))
    be.ctx = Ctx()
    P = projection_operators.ProjectionMatrix(geo, backend=be)
    A = np.zeros((N, N, N), np.float32)
    A[4:12, 4:12, 4:12] = 1.0
    B = np.zeros((N, N, N), np.float32)
    B[2:7, 3:9, 5:14] = 2.5
    pose = dict(alpha=0.01, beta=-0.02, phi=0.8, xyz_shift=np.array([0.5, 0., -0.7]), cor_shift=np.zeros(3))
    pA_ref, gA_ref = projection_operators.ProjectionMatrix(geo, backend=OracleBackend(geo)).projection_gradient(A, **pose)
    pB_ref, _ = projection_operators.ProjectionMatrix(geo, backend=OracleBackend(geo)).projection_gradient(B, **pose)
    reuse = lambda: [v for n, v in be.ctx.log if n == "reuse_staged_volume"][::2]      # the value set BEFORE each call
    P.pin_volume(A)
    p1, _ = P.projection_gradient(A, **pose)
    p2, _ = P.projection_gradient(A, **pose)
    assert reuse() == [0, 1] and np.array_equal(p1, pA_ref) and np.array_equal(p2, pA_ref)
    pB, _ = P.projection_gradient(B, **pose)             # another object while A is pinned
    assert np.array_equal(pB, pB_ref) and reuse()[-1] == 0 and not P._vol_staged
    p3, g3 = P.projection_gradient(A, **pose)            # A again: its own buffer, re-staged once, then reused
    p4, _ = P.projection_gradient(A, **pose)
    assert np.array_equal(p3, pA_ref) and np.array_equal(g3, gA_ref) and np.array_equal(p4, pA_ref)
    assert reuse()[-2:] == [0, 1] and P.volume_is_pinned(A) and not P.volume_is_pinned(B)
    # the same through the fused evaluation and its memo
    ao = alignment_functions.AlignmentUtilities(pA_ref.reshape(N, N), P, type("G", (), {"cor_shift": np.zeros(3)})())
    cA = ao.cost_and_gradient(A, (0.8, 0.01, -0.02), pose["xyz_shift"])[0]
    cB = ao.cost_and_gradient(B, (0.8, 0.01, -0.02), pose["xyz_shift"])[0]
    cA2 = ao.cost_and_gradient(A, (0.8, 0.01, -0.02), pose["xyz_shift"])[0]
    assert cA < 1e-10 and cB > 1.0 and cA2 == cA
    P.unpin_volume()
    assert not P.volume_is_pinned(A) and P._pin_dev is None


def test_in_place_edit_of_an_unpinned_volume_is_seen():
    """a phantom whose z = 0 plane and last voxel are empty, edited in place -- a fingerprint of strided
    samples cannot see the edit; every call must use the current contents."""
    N = 16
    geo = geom(1, N)
    P = projection_operators.ProjectionMatrix(geo, backend=OracleBackend(geo))
    rec = np.zeros((N, N, N), np.float32)
    rec[4:12, 4:12, 4:12] = 1.0
    pose = dict(alpha=0.01, beta=-0.02, phi=0.8, xyz_shift=np.array([0.5, 0., -0.7]), cor_shift=np.zeros(3))
    p1, g1 = P.projection_gradient(rec, **pose)
    rec *= 3.0                                  # in place: same pointer, shape, dtype; z = 0 plane and last voxel still 0
    p2, g2 = P.projection_gradient(rec, **pose)
    assert rel_max(p2, 3.0 * p1) < 1e-6 and rel_max(g2, 3.0 * g1) < 1e-6
    ao = alignment_functions.AlignmentUtilities(p1.reshape(N, N), P, type("G", (), {"cor_shift": np.zeros(3)})())
    c3 = ao.cost_and_gradient(rec, (0.8, 0.01, -0.02), pose["xyz_shift"])[0]
    rec[rec > 0] = 1.0                          # back to the original, in place
    c1 = ao.cost_and_gradient(rec, (0.8, 0.01, -0.02), pose["xyz_shift"])[0]
    assert c3 > 1.0 and c1 < 1e-8 * c3


def test_lbfgs_pose_recovery_and_gradient_descent_vs_reference_golden(shepp32):
    g, P, ao, args = _alignment_setup(shepp32)
    res = optimize.minimize(alignment_functions.cost_xzab, np.zeros(4), method="L-BFGS-B", jac=alignment_functions.gradient_xzab,
                            args=args, bounds=((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02)), options={"disp": False})
    assert np.allclose(res.x, g["true"], atol=2e-5)            # known-answer: the injected pose is recovered
    assert np.allclose(res.x, g["lbfgs_x"], atol=2e-5)
    gd = alignment_functions.gradient_descent
    x1, f1, stop1 = gd(np.zeros(4), alignment_functions.cost_xzab, alignment_functions.gradient_xzab, args=args + (None,),
                       options={"maxiter": 1})
    assert stop1 == int(g["gd1_stop"]) and np.allclose(x1, g["gd1_x"], rtol=1e-4, atol=1e-7) and np.isclose(f1, g["gd1_f"], rtol=1e-4)
    # five Armijo steps zig-zag between the translation and the (10^4 x stiffer) tilt directions: the iterate is
    # chaotic in the last float32 bit of the gradient, the cost reached is not
    x5, f5, stop5 = gd(np.zeros(4), alignment_functions.cost_xzab, alignment_functions.gradient_xzab, args=args + (None,),
                       options={"maxiter": 5})
    assert stop5 == int(g["gd_stop"]) and f5 < f1 and np.isclose(f5, g["gd_f"], rtol=0.2)


def test_reference_style_top_level_imports():
    """INTEGRATION.md level 1: with the package directory itself on sys.path, the reference's own import lines
    (`from utilities import ...`, `from recon import sirt`, examples/align_rigid.py:5-7) resolve to this package."""
    import subprocess
    import sys
    code = ("from utilities import geometry, projection_operators, alignment_functions, linear_operators, generate_phantom, rotations\n"
            "from recon import sirt, cgls, sirt_mpi, cgls_mpi\n"
            "from utilities.alignment_functions import cost_xzab, gradient_xzab\n"
            "import numpy as np\n"
            "g = geometry.Geometry(2, np.array([8, 8, 8]), np.ones(3), np.array([8, 8]), np.ones(2))\n"
            "assert projection_operators.ProjectionMatrix(g).precision is np.float32 and sirt.SIRT and cgls.CGLS\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tomography_alignment_amd"))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_batched_alignment_recovers_injected_poses(shepp32):
    """examples/align_rigid.py:40-52 for several projections at once: every projection's L-BFGS-B advances in lock
    step, one fused launch per round of evaluations."""
    from oracle import oracle as orc
    from tomography_alignment_amd import alignment
    from tomography_alignment_amd.comm import SingleComm
    n, N = 3, 32
    rng = np.random.default_rng(21)
    phi = np.array([0.5, 1.4, 2.3])
    true = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.deg2rad(rng.uniform(-1, 1, n)), np.deg2rad(rng.uniform(-1, 1, n))])
    og = orc.Geo(1, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2))
    b = np.array([orc.projection_gradient(og, shepp32, true[i, 2], true[i, 3], phi[i], np.array([true[i, 0], 0., true[i, 1]]), np.zeros(3))[0]
                  for i in range(n)])
    geo = geom(n, N)
    be = OracleBackend(geo)
    bounds = ((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02))
    res = alignment.align_projections(be, shepp32, b, phi, letters="xzab", bounds=bounds)
    assert np.allclose(res["x"], true, atol=5e-5) and np.all(res["fun"] < 1e-6)
    assert res["n_launch"] < res["n_eval"]                 # evaluations were batched
    assert res["n_launch"] == int(res["nfev"].max())       # lock step: one launch per round
    res2 = alignment.align_projections_sharded(SingleComm(), be, shepp32, b, phi, letters="xzab", bounds=bounds)
    assert np.allclose(res2["x"], res["x"], atol=1e-9)
    # round 4: one thread drives scipy's L-BFGS-B core for every projection (driver "batch"); round 3's thread-per-optimiser form
    # ("threads") stays as the fallback -- the iterates are scipy's own either way: identical x, fun, nfev
    assert alignment.batch_driver_available() and res["driver"] == "batch"
    res3 = alignment.align_projections(be, shepp32, b, phi, letters="xzab", bounds=bounds, driver="threads")
    assert res3["driver"] == "threads" and np.array_equal(res3["x"], res["x"]) and np.array_equal(res3["nfev"], res["nfev"]) and np.array_equal(res3["fun"], res["fun"])
    for i in range(n):                                     # ... and optimize.minimize itself on one projection at a time
        def fun(p, i=i):
            pose = np.array([[phi[i], p[2], p[3], p[0], 0., p[1], 0.]])
            c, g6 = be.cost_grad(pose, be.upload(shepp32), be.upload(b[i:i + 1]))
            return float(c[0]), g6[0][[0, 2, 4, 5]]
        r = optimize.minimize(fun, np.zeros(4), jac=True, method="L-BFGS-B", bounds=bounds, options={"disp": False})
        assert np.array_equal(r.x, res["x"][i]) and r.nfev == res["nfev"][i]


def test_batch_lbfgsb_driver_two_populations():
    """_lbfgsb_batch.minimize_many with >= 64 problems (two populations taking turns + the merged tail) against scipy.optimize.minimize,
    bit for bit; evaluation batches arrive from a helper thread, as in alignment.align_projections."""
    from concurrent.futures import ThreadPoolExecutor
    from tomography_alignment_amd import _lbfgsb_batch as lb
    assert lb.AVAILABLE and lb.self_test()
    rng = np.random.default_rng(0)
    n = 150
    A, c = rng.uniform(0.5, 3, (n, 4)), rng.uniform(-2, 2, (n, 4))

    def fg_one(i, p):
        d = p - c[i]
        return float(np.sum(A[i] * d ** 4 + d * d)), 4 * A[i] * d ** 3 + 2 * d
    bounds = ((-1, 1), (-1, None), (None, 0.5), (None, None))
    ref = [optimize.minimize(lambda p, i=i: fg_one(i, p), np.zeros(4), jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": 30}) for i in range(n)]
    sizes = []

    def fb(ids, X):
        sizes.append(len(ids))
        out = [fg_one(i, X[k]) for k, i in enumerate(ids)]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])
    with ThreadPoolExecutor(1) as ex:
        x, f, nf, st = lb.minimize_many(fb, np.zeros((n, 4)), bounds=bounds, options={"maxiter": 30, "disp": False}, overlap=ex.submit)
    assert all(np.array_equal(x[i], ref[i].x) and nf[i] == ref[i].nfev and f[i] == ref[i].fun and st[i] == ref[i].status for i in range(n))
    assert sum(sizes) == int(nf.sum()) and sizes[0] == 75 and sizes[1] == 75 and min(sizes) < 64      # two halves first, one population at the end


def test_bench_launcher_and_roofline_logic_without_a_gpu(tmp_path, monkeypatch):
    """bench.py on the CPU: (i) `--gpus 2` with no visible GPU fails cleanly before anything is started; (ii) the roofline picks
    the busiest unit from committed counters and never reports the algorithmic-HBM figure as the fraction when counters exist."""
    import json
    import subprocess
    import sys
    import ctypes
    n = ctypes.c_int(0)
    if _lib.load().tomo_device_count(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 2 and "2 GPUs requested, 0 visible" in out.stderr and out.stdout.strip() == ""
    # the command the driver's scaling run uses for its largest point: the launcher must get as far as counting GPUs
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 2 and "8 GPUs requested, 0 visible" in out.stderr and out.stdout.strip() == ""
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    live = _lib.kernel_source_hash()
    sqj = {"key": "K", "source": "t", "src_hash": live, "kernels": {"k_fwd_tile_flat": {"SQ_INSTS_VALU": 2.0e11, "SQ_INSTS_LDS": 6.0e10, "lds_bytes": 6.0e10 * 512,
                                                                                        "SQ_ACTIVE_INST_VALU": 1.5e11, "SQ_LDS_IDX_ACTIVE": 2.4e11}}}
    trj = {"key": "K", "source": "t", "src_hash": live, "kernels": {"k_fwd_tile_flat": {"hbm_bytes_per_launch": 4.0e11, "write_kb": 2.6e11 / 1024.0}}}
    json.dump(sqj, open(prof / "sq_counters.json", "w"))
    json.dump(trj, open(prof / "pmc_traffic.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r = bench.make_roofline("k_fwd_tile_flat", 500.0, 1.0, 4.4e12, "K", 1024.0 * 1024.0 ** 3, 4.0 * 1024 * 1024 ** 2)
    # useful work: 1024 angles x 1024^3 samples x 8 flop / 0.5 s = 17.6 TF of 157.3; atomics: 2.6e11 B / 0.5 s = 520 GB/s of 1300
    assert abs(r["useful_flop_frac"] - 8 * 1024.0 ** 4 / 0.5 / 1e12 / 157.3) < 1e-3 and abs(r["atomics_frac"] - 0.4) < 1e-3
    assert abs(r["atomics"]["amplification_vs_sinogram"] - 2.6e11 / (4.0 * 1024 ** 3)) < 0.1
    # LDS-array cycles 2.4e11 / 0.5 s = 480 G/s of the 614.4 G/s (256 CUs x 2.4 GHz) = 0.78; VALU 4 x 1.5e11 / 0.5 = 1200 of 2457.6 = 0.49
    assert r["bound"] == "lds" and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - 2.4e11 / 0.5 / 1e9 / 614.4) < 1e-3
    assert abs(r["utilisation"]["atomics"]["frac"] - 0.4) < 1e-3                       # the memory-side atomic unit is one of the candidates
    assert abs(r["utilisation"]["valu"]["frac"] - 4 * 1.5e11 / 0.5 / 1e9 / 2457.6) < 1e-3 and r["utilisation"]["hbm"]["frac"] < 0.2
    assert r["hbm_algorithmic"]["frac_of_hbm_peak"] > 1.0 and r["traffic"] == 4.0e11          # kept, labelled, not the roofline
    assert r["instruction_rates"]["lds_GBps"] > 0
    # both HBM fractions at the top of the block (VERDICT r3 #2a): the SURVEY 8(d) figure (4.4e12 B / 0.5 s / 8 TB/s = 1.1) and the counters' (0.1)
    assert abs(r["hbm_algorithmic_frac"] - 1.1) < 1e-3 and abs(r["hbm_counter_frac"] - 4.0e11 / 0.5 / 8e12) < 1e-3
    r2 = bench.make_roofline("k_fwd_tile_flat", 500.0, 1.0, 4.4e12, "other-workload")
    assert r2["bound"] == "hbm" and r2["counters"] is None and "no committed PMC counters" in r2["note"]
    assert abs(r2["hbm_algorithmic_frac"] - 1.1) < 1e-3 and r2["hbm_counter_frac"] is None
    # round 4: one file, several workloads (the headline, its dense-volume and tilted-pose legs)
    multi = lambda j, other: {"source": "t", "src_hash": live, "workloads": {"K": {"workload": "w", "kernels": j["kernels"]}, "K_D": {"workload": "d", "kernels": other}}}   # noqa: E731
    json.dump(multi(sqj, {"k_fwd_tile_flat": dict(sqj["kernels"]["k_fwd_tile_flat"], SQ_LDS_IDX_ACTIVE=1.2e11)}), open(prof / "sq_counters.json", "w"))
    json.dump(multi(trj, {"k_fwd_tile_flat": {"hbm_bytes_per_launch": 8.0e11, "write_kb": 2.6e11 / 1024.0}}), open(prof / "pmc_traffic.json", "w"))
    rm = bench.make_roofline("k_fwd_tile_flat", 500.0, 1.0, 4.4e12, "K", 1024.0 * 1024.0 ** 3, 4.0 * 1024 * 1024 ** 2)
    rd = bench.make_roofline("k_fwd_tile_flat", 500.0, 1.0, 4.4e12, "K_D", 1024.0 * 1024.0 ** 3, 4.0 * 1024 * 1024 ** 2)
    assert rm["bound"] == "lds" and abs(rm["frac"] - r["frac"]) < 1e-9 and rm["traffic"] == 4.0e11
    assert rd["traffic"] == 8.0e11 and abs(rd["utilisation"]["lds"]["frac"] - 1.2e11 / 0.5 / 1e9 / 614.4) < 1e-3 and "workload K_D" in rd["traffic_source"]
    assert bench.make_roofline("k_fwd_tile_flat", 500.0, 1.0, 4.4e12, "K_P")["counters"] is None
    # counters taken on OTHER kernel sources are refused, and the line says so
    sqj["src_hash"] = trj["src_hash"] = "0123456789abcdef"
    json.dump(sqj, open(prof / "sq_counters.json", "w"))
    json.dump(trj, open(prof / "pmc_traffic.json", "w"))
    r3 = bench.make_roofline("k_fwd_tile_flat", 500.0, 1.0, 4.4e12, "K", 1024.0 * 1024.0 ** 3, 4.0 * 1024 * 1024 ** 2)
    assert r3["counters"] is None and r3["traffic"] is None and r3["bound"] == "hbm" and len(r3["stale_counters_refused"]) == 2
    assert "atomics_frac" not in r3 and r3["useful_flop_frac"] > 0            # the live-time figure needs no counters
    assert len(live) == 16 and live == _lib.kernel_source_hash()


def test_bench_contract_line_stays_small_at_the_full_size_counters(monkeypatch):
    """round 5's stdout line was 25 KB (whole PMC dictionaries, per-outer tables) and the driver, which parses the last line
    out of a bounded tail of stdout, parsed nothing.  The stdout line is now bench.contract_line(record); this holds it under bench.LINE_LIMIT
    (4 KB) on (i) the very record that broke round 5 (profiles/round5_bench_1024.json, all legs, counter-backed rooflines) and (ii) rooflines
    rebuilt from the COMMITTED full-size counter files for every workload key they hold, and checks that every driver key survives."""
    import json
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "round5_bench_1024.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000                                        # the worst case on record
    sq = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))
    monkeypatch.setattr(_lib, "kernel_source_hash", lambda: sq["src_hash"])     # price against the committed counters whatever the sources are today
    N, n_proj = 1024, 1024
    records = [full]
    for key, w in sq["workloads"].items():
        rec = json.loads(json.dumps(full))
        if key.startswith("C5_"):
            per = {k: {"launches_per_pass": 1.0, "ms_per_pass": 60.0} for k in w["kernels"]}
            rec["alignment_gradient"]["roofline"] = bench.grad_roofline(per, 720 * (4.0 * 512 ** 3 + 4.0 * 512 ** 2 + 28.0), key, 720 * 512.0 ** 3)
            assert rec["alignment_gradient"]["roofline"]["counters"] is not None
        else:
            name = next(k for k in ("k_fwd_tile_flat", "k_fwd_tile") if k in w["kernels"])
            rec["roofline"] = bench.make_roofline(name, 200.0, 1.0, n_proj * (4.0 * N ** 3 + 4.0 * N * N), key, n_proj * float(N) ** 3, 4.0 * n_proj * N * N)
            rec["roofline"]["measured_d2d_copy_GBps"] = 5000.0
            assert rec["roofline"]["counters"] is not None and len(json.dumps(rec["roofline"])) > 1500
        records.append(rec)
    for rec in records:
        rec["detail_file"] = "gpurun_out/bench_detail.json"
        line = bench.contract_line(rec)
        assert len(line) < bench.LINE_LIMIT <= 4096 and "\n" not in line, len(line)
        d = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "roofline", "cpu_baseline", "value_dense_volume", "value_tilted_poses", "cgls_it_per_s", "alignment_gradient", "align_rigid_e2e", "detail"):
            assert k in d, k
        assert set(d["config"]) == {"workload", "sharding"} and d["value"] == rec["value"] and d["ms_per_step"] == rec["ms_per_step"]
        r = d["roofline"]
        for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_per_step", "traffic", "traffic_source",
                  "hbm_algorithmic_frac", "hbm_counter_frac"):
            assert k in r, k
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0 and "counters" not in r and "utilisation" not in r
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["all_cores"]["cores"] >= 1 and len(c["sample"]) <= 230
        assert c["reference"]["forward"]["s_per_angle"] > 0 and c["reference"]["gradient"]["s_per_eval"] > 0
        assert d["alignment_gradient"]["evals_per_sec"] > 0 and d["align_rigid_e2e"]["wall_s"] > 0
        assert sum(d["kernel_ms_per_step"].values()) <= d["ms_per_step"] * 1.02           # kernel time <= step time, checkable from the line
    # nothing measured is dropped: the record itself goes to bench_detail.json / stderr (bench.emit)
    import io
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        rfd, wfd = os.pipe()
        err = io.StringIO()
        monkeypatch.setattr(sys, "stderr", err)
        bench.emit(dict(full), wfd, os.path.join(td, "detail.json"))
        os.close(wfd)
        got = os.read(rfd, 1 << 16).decode()
        os.close(rfd)
        assert got.count("\n") == 1 and len(got) < bench.LINE_LIMIT and json.loads(got)["detail"].endswith("detail.json")
        back = json.load(open(os.path.join(td, "detail.json")))
        assert back["alignment_gradient"]["roofline"]["counters"] == full["alignment_gradient"]["roofline"]["counters"] and back["kernels"] == full["kernels"]
        assert "SQ_INSTS_VALU" in err.getvalue()


def test_samples_per_ray_match_numpy_rounding():
    """n = int(|r0| / step) (utilities/ray_voxel_utilities.py:88) hangs on the rounding of a length that is an integer in exact
    arithmetic.  The library's host code (csrc/tomo_raycore.h, built here as a plain C++ program) rounds its 3-term inner products in
    a DOCUMENTED way -- one rounded product, two FMAs in ascending k, the norm from three rounded squares added in order -- which is
    what np.dot / np.linalg.norm do on an FMA machine.  Primary check (host-independent): |r0| bit for bit and n for every pose
    against an exact-arithmetic emulation of that rounding.  Secondary check (depends on the host's BLAS): the same
    against this host's numpy, only where a probe shows that its np.dot rounds the fused way."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("n_check", os.path.join(ROOT, "tools", "n_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(400, against="emulation") == 0
    if mod.numpy_dot_is_fused():
        assert mod.main(600, against="numpy") == 0       # n differed for 8 % of the poses before the rounding was matched
    else:
        print("this host's np.dot does not round 3-term products the fused way: numpy parity of n not checked here")


def test_reference_callers_run_unchanged_on_this_operator():
    """VERDICT r3 "missing" #2: the reference's own recon/sirt.py::SIRT and utilities/alignment_functions.py, imported UNMODIFIED, with
    `utilities.projection_operators` swapped for this package's ProjectionMatrix (INTEGRATION.md level 2; CPU stand-in backend), reproduce
    the goldens the unswapped reference wrote (G5 rec / rms_error, G6 cost / gradient / L-BFGS-B x).  Authoring container only: skipped
    where the reference tree is absent (the GPU box) -- profiles/round4_ref_callers_unchanged.log is the committed output."""
    import subprocess
    if not os.path.isdir("/root/reference/recon"):
        pytest.skip("reference tree not present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_callers_unchanged.py")], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "run unchanged on this package's operator" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_f2py_twin_module_validates_and_fails_loudly_without_a_gpu():
    """`src.ray_wt_grad` (the f2py module's name and signatures, INTEGRATION.md level 2 1/2): argument validation happens on the host, and
    without a GPU the first real call raises TomoError -- there is no CPU fallback behind this surface either."""
    from tomography_alignment_amd.src import ray_wt_grad
    fp = np.zeros((3, 4, 5), np.int32)
    wf = np.zeros((3, 4, 5))
    with pytest.raises(ValueError):
        ray_wt_grad.trilinear_ray_sparse(fp, wf[:, :3], 8, 8, 8, 4, 5)
    with pytest.raises(ValueError):
        ray_wt_grad.trilinear_ray_interp(fp, wf, 8, 8, 8, 4, 5, np.zeros(512), np.zeros((4, 4)), np.zeros((9, 3, 4)))
    with pytest.raises(ValueError):
        ray_wt_grad.trilinear_ray_interp(fp, wf, 8, 8, 8, 4, 5, np.zeros(100), np.zeros((4, 5)), np.zeros((9, 3, 4)))
    # the other f2py module (round 6): src.vox_wt_grad
    from tomography_alignment_amd.src import vox_wt_grad
    import inspect
    assert list(inspect.signature(vox_wt_grad.bilinear_sparse).parameters) == ["n_vox", "floor_x", "floor_z", "alpha_x", "alpha_z", "ndim_x", "ndim_z"]
    assert list(inspect.signature(vox_wt_grad.bilinear_vox_interp).parameters) == ["n_vox", "floor_x", "floor_z", "alpha_x", "alpha_z", "rec", "ndim_x", "ndim_z",
                                                                                   "der_points"]
    fx, ax = np.zeros(10, np.int32), np.zeros(10, np.float32)
    with pytest.raises(ValueError):
        vox_wt_grad.bilinear_sparse(11, fx, fx, ax, ax, 4, 4)                                             # arrays shorter than n_vox
    with pytest.raises(ValueError):
        vox_wt_grad.bilinear_vox_interp(10, fx, fx, ax, ax, ax, 4, 4, np.zeros((6, 3, 9), np.float32))    # der_points too short
    with pytest.raises(ValueError):
        vox_wt_grad.bilinear_vox_interp(10, fx, fx, ax, ax, ax, 0, 4, np.zeros((6, 3, 10), np.float32))
    n = ctypes.c_int(0)
    if _lib.load().tomo_device_count(ctypes.byref(n)) != 0 or n.value == 0:
        with pytest.raises(_lib.TomoError):
            ray_wt_grad.trilinear_ray_sparse(fp, wf, 8, 8, 8, 4, 5)
        with pytest.raises(_lib.TomoError):
            vox_wt_grad.bilinear_sparse(10, fx, fx, ax, ax, 4, 4)
        with pytest.raises(_lib.TomoError):
            vox_wt_grad.bilinear_vox_interp(10, fx, fx, ax, ax, ax, 4, 4, np.zeros((6, 3, 10), np.float32))


def test_reference_python_imports_over_the_package_src():
    """INTEGRATION.md level 2 1/2 (VERDICT r5 missing 3): ALL of the reference's Python, only `src` replaced.  The reference's
    utilities/projection_operators.py:7-8 imports utilities.voxel_utilities, which does `from src import vox_wt_grad` -- an ImportError until
    round 6.  tools/ref_over_package_src.py (authoring container only) imports the reference's utilities over this package's `src`, checks
    that every reference module resolved to /root/reference and both `src` modules to this package, and that the reference's callers reach the
    twins (a call without a GPU raises this package's TomoError from inside the reference's own forward_sparse / forward_proj_grad)."""
    import subprocess
    if not os.path.isdir("/root/reference/utilities"):
        pytest.skip("reference tree not present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_over_package_src.py")], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "RESULT: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_ray_voxel_utilities_mirror(shepp32):
    """utilities/ray_voxel_utilities.py under the reference's function names (round 4): the two numpy helpers against the oracle's
    restatement, `forward_sparse` / `forward_proj_grad` (CPU stand-in backend here; the GPU twin is tests/test_gpu_parity.py) against the
    reference's goldens G1 b (assembled CSR) and G3 (projection_gradient), and the geometry is NOT shifted in place."""
    import copy
    from oracle import oracle as orc
    from tomography_alignment_amd.utilities import ray_voxel_utilities as rvu
    rng = np.random.default_rng(4)
    x = rng.standard_normal((3, 7))
    t = rng.standard_normal(3)
    assert np.allclose(rvu.transform_points(x, 0.01, -0.02, 1.1, t), orc.transform_points(x, 0.01, -0.02, 1.1, t), rtol=0, atol=1e-15)
    assert np.allclose(rvu.derivative_ray_points(x, np.array([0., 16., 0.]), 0.01, -0.02, 1.1, t),
                       orc.derivative_ray_points(x, np.array([0., 16., 0.]), 0.01, -0.02, 1.1, t), rtol=0, atol=1e-13)
    g1 = golden("g1_operator")
    N, n_proj = 8, len(g1["b_phi"])
    rows, cols, vals = [], [], []
    for ip in range(n_proj):
        geo = geom(1, N)
        geo.cor_shift = g1["b_cor"][ip]                      # as utilities/projection_operators.py:101-102 hands it over
        before = copy.deepcopy(geo.source_centers)
        dat, det, wts = rvu.forward_sparse(geo, g1["b_alpha"][ip], g1["b_beta"][ip], g1["b_phi"][ip], g1["b_xyz"][ip], backend=OracleBackend(geo))
        assert np.array_equal(geo.source_centers, before) and dat.dtype == np.int32 and wts.dtype == np.float64
        rows.append(det.astype(np.int64) + ip * N * N)
        cols.append(dat.astype(np.int64))
        vals.append(wts)
    A = sparse.csr_matrix(sparse.coo_matrix((np.concatenate(vals).astype(np.float32), (np.concatenate(rows), np.concatenate(cols))), shape=tuple(g1["b_shape"])))
    A.sum_duplicates()
    A.sort_indices()
    assert np.array_equal(A.indptr, g1["b_indptr"]) and np.array_equal(A.indices, g1["b_indices"]) and rel_max(A.data, g1["b_data"]) < 1e-6
    # ProjectionMatrix._forward_ray (reference utilities/projection_operators.py:95-110): the lists its projection_matrix concatenates
    geo = Geometry(n_proj, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2), cor_shift=g1["b_cor"])
    P = projection_operators.ProjectionMatrix(geo, backend=OracleBackend(geo))
    P.projection_matrix(alpha=g1["b_alpha"], beta=g1["b_beta"], phi=g1["b_phi"], xyz_shift=g1["b_xyz"])
    w, di, da = P._forward_ray()
    A2 = sparse.csr_matrix(sparse.coo_matrix((np.concatenate(w), (np.concatenate(di), np.concatenate(da))), shape=tuple(g1["b_shape"])))
    A2.sum_duplicates()
    A2.sort_indices()
    assert w[0].dtype == np.float32 and np.array_equal(A2.indices, g1["b_indices"]) and rel_max(A2.data, g1["b_data"]) < 1e-6
    g3 = golden("g3_proj_grad")
    for i in range(3):
        geo = geom(1, 32)
        geo.cor_shift = g3["cor"][i]
        p, gr = rvu.forward_proj_grad(geo, g3["alpha"][i], g3["beta"][i], g3["phi"][i], g3["xyz"][i], shepp32, backend=OracleBackend(geo))
        assert p.dtype == np.float64 and gr.shape == (6, 1024) and rel_max(p, g3["proj"][i]) < 1e-5 and rel_max(gr, g3["grad"][i]) < 1e-5


def test_sirt_regularized_gradient_descent_vs_reference_golden_g12(shepp32):
    """recon/sirt.py::SIRT.run_regularized_gradient_descent + my_f / my_fp (reference :109-197; round 4) on the CPU stand-in backend against
    golden G12 (the reference's class on its own CSR): scipy's strong-Wolfe line search runs on this package's operator."""
    from tomography_alignment_amd.recon import sirt as sirt_mod
    g5, g = golden("g5_sirt"), golden("g12_sirt_regularized_gd")
    geo = geom(16, 32)
    angles = np.array([g5["phi"], g5["alpha"], g5["beta"]]).T
    for tag, gt in (("a", None), ("b", shepp32)):
        opts = {"_backend": OracleBackend(geo)}
        if gt is not None:
            opts["ground_truth"] = gt.copy()
        s = sirt.SIRT(geo, g5["b"].copy(), angles, g5["xyz"], options=opts)
        rec, err = s.run_regularized_gradient_descent(niter=int(g["nit_" + tag]), reg_param=float(g["reg_" + tag]), positivity=bool(g["pos_" + tag]))
        assert rec.shape == (32, 32, 32) and len(err) == len(g["err_" + tag])
        assert rel_max(rec, g["rec_" + tag]) < 2e-5 and np.allclose(err, g["err_" + tag], rtol=2e-5)
    assert abs(sirt_mod.my_f(g["f_x"], s.proj_mat, g5["b"], 0.7) / float(g["my_f"]) - 1) < 1e-6
    assert rel_max(sirt_mod.my_fp(g["f_x"], s.proj_mat, g5["b"], 0.7), g["my_fp"]) < 1e-5
