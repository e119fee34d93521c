"""GPU tests of the layers above the projectors: device-resident SIRT / CGLS, the alignment API, solver vector
kernels, the device phantom, event timing, and RCCL bring-up (1 rank) -- all through the C-ABI."""
import copy
import os
import time

import numpy as np
import pytest
from scipy import optimize

from conftest import golden, rel_max

pytestmark = pytest.mark.gpu


def geom(n_proj, N, **kw):
    from tomography_alignment_amd.utilities.geometry import Geometry
    return Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2), **kw)


@pytest.mark.parametrize("tag,positivity,use_gt", [("plain", False, False), ("pos_gt", True, True)])
def test_sirt_vs_reference_golden(shepp32, tag, positivity, use_gt):
    from tomography_alignment_amd.recon import sirt
    g = golden("g5_sirt")
    geo = geom(16, 32)
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    opts = {"ground_truth": shepp32.copy()} if use_gt else {}
    s = sirt.SIRT(geo, g["b"].copy(), angles, g["xyz"], options=opts)
    # W = 1/(A.1) blows up for rays that graze a corner (row sum ~1e-3): compare the row / column sums themselves
    inv = lambda v: np.divide(1.0, v, out=np.zeros_like(v), where=v != 0)    # noqa: E731
    assert np.array_equal(s.W == 0, g["W"] == 0) and rel_max(inv(s.W), inv(g["W"])) < 1e-5
    assert np.array_equal(s.V == 0, g["V"] == 0) and rel_max(inv(s.V), inv(g["V"])) < 1e-5
    rec, err = s.run_main_iteration(niter=10, positivity=positivity)
    assert rec.shape == (32, 32, 32)
    e_rec, e_err = rel_max(rec, g["rec_" + tag]), float(np.max(np.abs(err - g["err_" + tag]) / g["err_" + tag]))
    print("SIRT x10 vs the reference (%s): rec rel-max %.2e, rms_error rel %.2e" % (tag, e_rec, e_err))
    # measured 3-4e-7; the conditioning of 10 iterations (the reference's own iterate moves by 5x an operator perturbation,
    # tests/test_oracle_golden.py::test_sirt_sensitivity_to_operator_rounding) leaves ample room inside 1e-5
    assert e_rec < 1e-5 and e_err < 1e-5


def test_sharded_sirt_world_of_one_matches_plain(shepp32):
    from tomography_alignment_amd.recon import sirt_mpi
    from tomography_alignment_amd.comm import SingleComm
    g = golden("g5_sirt")
    geo = geom(16, 32)
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    s = sirt_mpi.SIRT(SingleComm(), geo, g["b"].copy(), angles, g["xyz"], options={})
    rec, err = s.run_main_iteration(niter=10)
    assert rel_max(rec, g["rec_plain"]) < 1e-5


def test_cgls_vs_restated_reference(shepp32):
    """The device-resident CGLS against the oracle's restatement (oracle/oracle.py::Cgls, itself pinned to the reference's class by
    golden G11 since round 4; the direct comparison with G11 is the next test).  8 iterations are held to 1e-5 (measured 2e-7, printed)."""
    from oracle import oracle as orc
    from tomography_alignment_amd.recon import cgls
    g = golden("g5_sirt")
    geo = geom(16, 32)
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    og = orc.Geo(16, np.array([32] * 3), np.ones(3), np.array([32, 32]), np.ones(2))
    kw = dict(alpha=g["alpha"], beta=g["beta"], phi=g["phi"], xyz_shift=g["xyz"])
    want, want_err = orc.cgls(lambda x: orc.forward(og, x, **kw).astype(np.float32).ravel(),
                              lambda y: orc.adjoint(og, y, **kw).astype(np.float32), 32 ** 3, g["b"], 8)
    rec, err = cgls.CGLS(geo, g["b"].copy(), angles, g["xyz"]).run_main_iteration(niter=8)
    print("CGLS x8 vs the oracle's restatement: rec rel-max %.2e, rms rel %.2e" % (rel_max(rec, want), float(np.max(np.abs(err - want_err) / want_err))))
    assert rel_max(rec, want) < 1e-5 and np.allclose(err, want_err, rtol=1e-5)


def test_cgls_vs_reference_golden_g11(shepp32, capsys):
    """Device-resident CGLS against the REFERENCE'S OWN CLASS (golden G11, round 4: recon/cgls.py::CGLS executed from the reference tree
    on the CSR its projection_matrix returns; tests/golden/make_golden.py::g11): a / b -- 10 iterations on G5's sinogram without /
    with a ground truth at 1e-5; c -- the operator swapped under the solver after 3 iterations: the re-initialisation rule fires at
    iteration 6 and the run continues (recon/cgls.py:60-70); d -- the rise comes at k = 1: the reference quits with one rms value."""
    from tomography_alignment_amd.recon import cgls
    g5, g = golden("g5_sirt"), golden("g11_cgls")
    geo = geom(16, 32)
    angles = np.array([g5["phi"], g5["alpha"], g5["beta"]]).T
    for tag, gt in (("a", None), ("b", shepp32)):
        opts = {} if gt is None else {"ground_truth": gt.copy()}
        rec, err = cgls.CGLS(geo, g5["b"].copy(), angles, g5["xyz"], options=opts).run_main_iteration(niter=10)
        e, er = rel_max(rec, g["rec_" + tag]), float(np.max(np.abs(err - g["err_" + tag]) / g["err_" + tag]))
        with capsys.disabled():
            print("CGLS x10 vs the reference's class (G11 %s): rec rel-max %.2e, rms rel %.2e" % (tag, e, er))
        assert e < 1e-5 and er < 1e-5
    for tag in ("c", "d"):
        geo = geom(6, 16)
        ang = np.array([g[tag + "_phi"], g[tag + "_alpha"], g[tag + "_beta"]]).T
        capsys.readouterr()
        c = cgls.CGLS(geo, g[tag + "_b"].copy(), ang, g[tag + "_xyz"])
        rec, err1 = c.run_main_iteration(niter=int(g[tag + "_first"]))
        c.xyz_shift = g[tag + "_xyz2"]
        c.proj_mat = c.f_proj_obj.projection_matrix(phi=ang[:, 0], alpha=ang[:, 1], beta=ang[:, 2], xyz_shift=g[tag + "_xyz2"])
        rec, err = c.run_main_iteration(niter=12)
        said = capsys.readouterr().out
        assert np.allclose(err1, g["err1_" + tag], rtol=1e-5)
        assert len(err) == len(g["err_" + tag]) and said.count("reinitializing") == int(g[tag + "_reinit_lines"]) and int("quitting" in said) == int(g[tag + "_quit"])
        e = rel_max(rec, g["rec_" + tag])
        er = float(np.max(np.abs(err - g["err_" + tag]) / g["err_" + tag]))
        with capsys.disabled():
            print("CGLS re-initialisation (G11 %s): %d iterations, rec rel-max %.2e, rms rel %.2e" % (tag, len(err), e, er))
        assert e < 2e-5 and er < 2e-5          # the swap makes the iteration ill-conditioned on purpose: an operator rounding is amplified (measured, printed)


def test_linear_operators_module(shepp32):
    from tomography_alignment_amd.utilities import linear_operators
    g = golden("g2_fwd_adj")
    op = linear_operators.LinearOperator(geom(6, 32))
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    assert rel_max(op.project(shepp32, 6, angles, g["xyz"]).ravel(), g["Ax"]) < 1e-5
    assert rel_max(op.backproject(g["y"], 6, angles, g["xyz"]), g["ATy"]) < 1e-5


def test_alignment_api_vs_reference_golden(shepp32, capsys):
    """utilities/alignment_functions.py:113-485 on the GPU backend against the reference's own outputs (golden G6): ALL eleven
    cost_* / gradient_* pairs, their scale_factor and return_vector modes, the finite-difference checkers, gradient_descent (:40-110)
    and the L-BFGS-B recovery.  Scalars and gradients at 1e-5 (measured errors printed)."""
    from tomography_alignment_amd.utilities import projection_operators, alignment_functions as af
    g = golden("g6_alignment")
    geo = geom(1, 32)
    this_geo = copy.copy(geo)
    this_geo.cor_shift = geo.cor_shift[0]
    P = projection_operators.ProjectionMatrix(geo)
    ao = af.AlignmentUtilities(g["b"].reshape(32, 32), P, this_geo)
    args = (ao, shepp32, np.array([float(g["phi0"]), 0., 0.]), np.zeros(3))
    TOL = 1e-5
    errs = {}

    def chk(name, got, want):
        e = rel_max(np.atleast_1d(got), np.atleast_1d(want))
        errs[name] = e
        assert e < TOL, (name, e, got, want)
    for tag in ("zero", "gen"):
        p = g["p_" + tag]
        chk("cost_xzab@" + tag, af.cost_xzab(p, *args), g["cost_xzab_" + tag])
        chk("gradient_xzab@" + tag, af.gradient_xzab(p, *args), g["grad_xzab_" + tag])
        p5 = np.array([p[0], p[1], 0.003, p[2], p[3]])
        chk("cost_xzpab@" + tag, af.cost_xzpab(p5, *args), g["cost_xzpab_" + tag])
        chk("gradient_xzpab@" + tag, af.gradient_xzpab(p5, *args), g["grad_xzpab_" + tag])
    pg = g["p_gen"]
    chk("gradient_xzab scale_factor", af.gradient_xzab(pg, *args, scale_factor=np.array([1.0, 2.0, 50.0, 25.0])), g["grad_xzab_scaled"])
    chk("gradient_xzab return_vector", af.gradient_xzab(pg, *args, return_vector=True), g["grad_xzab_vec"])
    chk("cost_xzab return_vector", af.cost_xzab(pg, *args, return_vector=True), g["cost_xzab_vec"])
    for nm, p in (("xz", [0.4, -0.7]), ("x", [0.4]), ("z", [-0.7]), ("ab", [0.004, -0.006]), ("a", [0.004]), ("b", [-0.006]),
                  ("xzb", [0.4, -0.7, -0.006])):            # with xzpab and xzab above: the reference's eleven pairs
        p = np.array(p)
        chk("cost_" + nm, getattr(af, "cost_" + nm)(p, *args), g["cost_" + nm])
        chk("gradient_" + nm, getattr(af, "gradient_" + nm)(p, *args), g["grad_" + nm])
    # the reference's finite-difference checkers agree with its analytic gradient to FD accuracy only (trilinear interpolation is C0)
    assert rel_max(af.gradient_xz_fd(np.array([0.4, -0.7]), *args), g["grad_xz"]) < 5e-2
    res = optimize.minimize(af.cost_xzab, np.zeros(4), method="L-BFGS-B", jac=af.gradient_xzab, args=args,
                            bounds=((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02)), options={"disp": False})
    assert np.allclose(res.x, g["true"], atol=5e-5)        # the injected (tx, tz, alpha, beta) is recovered
    assert np.allclose(res.x, g["lbfgs_x"], atol=5e-5) and res.fun < 1e-6
    # gradient_descent: one Armijo step is reproducible; five zig-zag between the translation and the (10^4 x stiffer) tilt directions
    # and are chaotic in the last bit of the gradient -- the stop code and the cost level reached are what the reference's run pins
    x1, f1, stop1 = af.gradient_descent(np.zeros(4), af.cost_xzab, af.gradient_xzab, args=args + (None,), options={"maxiter": 1})
    assert stop1 == int(g["gd1_stop"]) and np.allclose(x1, g["gd1_x"], rtol=1e-4, atol=1e-7)
    chk("gradient_descent maxiter=1 cost", f1, g["gd1_f"])
    x5, f5, stop5 = af.gradient_descent(np.zeros(4), af.cost_xzab, af.gradient_xzab, args=args + (None,), options={"maxiter": 5})
    # (the reference's run ends at 187.6, the float64 oracle backend at 214.3, this backend at ~296: same descent, different zig-zag)
    assert stop5 == int(g["gd_stop"]) and f5 < 0.5 * f1 and 1.0 / 3 < f5 / float(g["gd_f"]) < 3.0, (f1, f5, g["gd_f"])
    with capsys.disabled():
        print("\n[alignment API vs reference G6, GPU backend] worst %.1e (%s); " % (max(errs.values()), max(errs, key=errs.get)) +
              ", ".join("%s %.1e" % kv for kv in sorted(errs.items())))


def test_vector_kernels_phantom_timer_profile():
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN, shepp3d
    geo = geom(2, 32)
    be = HipBackend(geo)
    rng = np.random.default_rng(0)
    n = 100003
    a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    da, db, dc = be.upload(a), be.upload(b), be.empty(n)
    assert np.isclose(be.dot(da, db), np.dot(a.astype(np.float64), b.astype(np.float64)), rtol=1e-10)
    assert np.isclose(be.diff_sumsq(da, db), np.sum((a.astype(np.float64) - b) ** 2), rtol=1e-6)
    be.sub(dc, da, db)
    assert np.array_equal(dc.download(), a - b)
    be.axpy(dc, da, 0.5)
    assert np.allclose(dc.download(), (a - b) + np.float32(0.5) * a, rtol=1e-6, atol=1e-6)
    be.xpay(dc, db, 2.0)
    w = a.copy()
    w[::7] = 0.0
    dw = be.upload(w)
    be.recip_guard(dw)
    got = dw.download()
    assert np.all(got[::7] == 0.0) and np.allclose(got[1::7], 1.0 / w[1::7], rtol=1e-6)
    dw = be.upload(np.abs(w) * 1e-7)
    be.recip_guard(dw, 1e-8)
    assert np.all(dw.download()[np.abs(w) * 1e-7 < 1e-8] == 0.0)
    s = be.residual_scale(da, db, None, dc)
    assert np.isclose(s, np.sum((a.astype(np.float64) - b) ** 2), rtol=1e-6) and np.array_equal(dc.download(), a - b)
    vol = be.phantom(be.empty(32 ** 3), (32, 32, 32), SHEPP_LOGAN).download().reshape(32, 32, 32)
    assert np.mean(vol != shepp3d(32)) < 1e-3 and rel_max(vol, golden("g7_phantom")["shepp32"]) <= 1.0
    be.ctx.profile_reset()
    be.ctx.profile_enable(True)
    be.ctx.timer_start()
    be.dot(da, db)
    ms = be.ctx.timer_stop()
    be.ctx.profile_enable(False)
    n_launch, tot = be.ctx.profile_get("k_dot")
    assert n_launch == 1 and 0.0 < tot <= ms + 1.0


def test_rccl_single_rank_allreduce():
    import os
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")      # single node: bootstrap over loopback (no NIC probing)
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.comm import RcclComm
    ctx = _lib.Context(0)
    comm = RcclComm(ctx, 0, 1, RcclComm.unique_id(ctx.lib))
    x = ctx.to_device(np.arange(1000, dtype=np.float32))
    comm.allreduce_sum_(x)
    ctx.sync()
    assert np.array_equal(x.download(), np.arange(1000, dtype=np.float32))
    assert comm.allreduce_scalar(2.5) == 2.5 and comm.allreduce_max(-1.0) == -1.0
    comm.close()


def test_example_drivers_recover_misalignment():
    """generate_data -> align_rigid: the known-answer smoke test the reference's examples amount to."""
    from tomography_alignment_amd.examples import generate_data, align_rigid
    data = generate_data.make(size=32, n_proj=24, seed=3)
    assert data["projections"].shape == (24, 32, 32)
    rec, a, b, xyz, hist = align_rigid.run(data, n_outer=3, sirt_iters=40, verbose=False)
    doing_nothing = np.abs(data["xyz"][:, [0, 2]]).mean()                          # ~1 px of injected jitter
    assert hist[-1]["shift_err_px"] < 0.3 * doing_nothing and hist[-1]["shift_err_px"] <= hist[0]["shift_err_px"]
    assert hist[0]["launches"] < hist[0]["evals"]                                  # evaluations were batched


def test_example_drivers_recover_the_tilts():
    """alpha, beta recovery was asserted nowhere.  The reference's own example sizes -- 64^3, 90 angles, alpha, beta ~ +-1 deg,
    tx, tz ~ +-2 px (examples/generate_data.py:17-23), bounds +-3 px / +-0.02 rad (examples/align_rigid.py:48) -- six outer iterations of 50 SIRT
    iterations: the tilt error must end below HALF of the injected mean (measured: 0.5 deg injected -> 1.07 after the first pass, which aligns
    against a reconstruction blurred by the very misalignment, -> 0.13 after the sixth; profiles/round6_config5_convergence.md has the 16-iteration
    curve down to 0.013 deg and the 512^3 x 720 run), fall from the second pass on, and the shifts must be recovered to a twentieth of a pixel."""
    from tomography_alignment_amd.examples import generate_data, align_rigid
    data = generate_data.make(size=64, n_proj=90, seed=3)
    rec, a, b, xyz, hist = align_rigid.run(data, n_outer=6, sirt_iters=50, verbose=False)
    tilt0 = float(np.rad2deg(np.abs(np.column_stack([data["alpha"], data["beta"]])).mean()))
    shift0 = float(np.abs(data["xyz"][:, [0, 2]]).mean())
    tilt = [h["tilt_err_deg"] for h in hist]
    print("tilt recovery at 64^3 x 90: injected mean %.3f deg -> %s; shifts %.3f px -> %.3f" % (tilt0, ", ".join("%.3f" % t for t in tilt), shift0, hist[-1]["shift_err_px"]))
    assert 0.4 < tilt0 < 0.6 and 0.8 < shift0 < 1.2
    assert tilt[-1] < 0.5 * tilt0, tilt
    assert all(tilt[i + 1] < tilt[i] for i in range(1, len(tilt) - 1)), tilt
    assert hist[-1]["shift_err_px"] < 0.05 * shift0 + 0.03 and hist[-1]["rmse"] < 0.6 * hist[0]["rmse"]
    # the recovered tilts themselves, projection by projection: three quarters of them within 0.25 deg of the truth
    err = np.rad2deg(np.abs(np.column_stack([a - data["alpha"], b - data["beta"]])))
    assert np.mean(err < 0.25) > 0.75, float(np.mean(err < 0.25))


def test_batched_alignment_gpu(shepp32):
    from oracle import oracle as orc
    from tomography_alignment_amd import alignment
    from tomography_alignment_amd.backend import HipBackend
    n, N = 5, 32
    rng = np.random.default_rng(22)
    phi = np.linspace(0.3, 2.8, n)
    true = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.deg2rad(rng.uniform(-1, 1, n)), np.deg2rad(rng.uniform(-1, 1, n))])
    og = orc.Geo(1, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2))
    b = np.array([orc.projection_gradient(og, shepp32, true[i, 2], true[i, 3], phi[i], np.array([true[i, 0], 0., true[i, 1]]), np.zeros(3))[0]
                  for i in range(n)])
    res = alignment.align_projections(HipBackend(geom(n, N)), shepp32, b, phi, letters="xzab",
                                      bounds=((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02)))
    assert np.allclose(res["x"], true, atol=1e-4) and res["n_launch"] < res["n_eval"]


def test_pipelined_allreduce_path_matches_plain(shepp32):
    """The x-slab back-projection + asynchronous all-reduce sequence of the sharded SIRT, driven through a real
    1-rank RCCL communicator (same streams / events as N ranks), against the plain sequence."""
    import os
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.comm import RcclComm
    from tomography_alignment_amd.recon import sirt_mpi
    g = golden("g5_sirt")
    N = 32
    geo = geom(16, N)
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    untilted = np.array([g["phi"], 0 * g["alpha"], 0 * g["beta"]]).T
    ctx = _lib.Context(0)
    comm = RcclComm(ctx, 0, 1, RcclComm.unique_id(ctx.lib))
    for ang in (angles, untilted):
        res = {}
        for force in (False, True):
            comm.force_pipeline = force
            s = sirt_mpi.SIRT(comm, geo, g["b"].copy(), ang, g["xyz"], options={"_backend": HipBackend(geo, ctx=ctx)})
            s.n_pipeline_slabs = 3
            res[force] = s.run_main_iteration(niter=6)
        assert rel_max(res[True][0], res[False][0]) < 2e-6 and np.allclose(res[True][1], res[False][1], rtol=1e-6)
    be = HipBackend(geo, ctx=ctx)
    assert be.xslab_info() == (3, 16)
    # (The control of the poison hook -- an un-waited read MUST see the doubled buffer -- and the mutation "a wait goes missing" run in a
    #  fresh child process, test_stream_overlap_waits_have_teeth below: in this process, late in the suite, the two streams may share one
    #  hardware queue, where nothing can race.  Here the poisoned runs only have to equal the plain sequence.)
    # round 3: the NEXT iteration's forward projection is made slab by slab behind the update (tomo_forward_xslab, tomo_comm_wait_next).
    # A ragged volume with 6 tile columns, flat and tilted poses, positivity and a ground truth (the error sum accumulates over the
    # slabs on the device); per iteration every slab's all-reduce is waited for once and iterations 2.. launch no whole forward.
    from oracle import oracle as orc
    # (200 planes, the object in planes 70 .. 149: the flat forward's live-block lists, empty sinogram planes and the gather
    #  back-projection's z chunks that end at once all take part in the slab calls)
    shape, ndet, n_proj = (80, 40, 200), (72, 210), 12
    rng = np.random.default_rng(11)
    x = np.zeros(shape, np.float32)
    x[10:70, 6:34, 70:150] = rng.uniform(0.2, 1.0, (60, 28, 80)).astype(np.float32)
    phi = np.linspace(0.05, np.pi - 0.05, n_proj)
    from tomography_alignment_amd.utilities.geometry import Geometry
    geo2 = Geometry(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2))
    og = orc.Geo(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2))
    for tilt in (0.0, 1.0):
        alpha, beta = np.deg2rad(tilt * rng.uniform(-1.5, 1.5, n_proj)), np.deg2rad(tilt * rng.uniform(-1.5, 1.5, n_proj))
        xyz = np.zeros((n_proj, 3))
        xyz[:, 0], xyz[:, 2] = rng.uniform(-2, 2, n_proj), rng.uniform(-2, 2, n_proj)
        b = orc.forward(og, x, alpha=alpha, beta=beta, phi=phi, xyz_shift=xyz).astype(np.float32)
        ang = np.array([phi, alpha, beta]).T
        res = {}
        # (force the slab pipeline, slabs, shard the update: reduce-scatter -> update of the own piece -> all-gather [round 4] or
        #  all-reduce + whole update [round 3], poison: every collective preceded by the doubling / idling / halving of its buffer)
        for force, slabs, shard, poison in ((False, 8, True, 0), (True, 8, True, 0), (True, 8, True, 300), (True, 8, False, 300), (True, 4, True, 300),
                                            (True, 2, True, 0), (True, 2, False, 0)):
            comm.force_pipeline = force
            s = sirt_mpi.SIRT(comm, geo2, b.copy(), ang, xyz, options={"_backend": HipBackend(geo2, ctx=ctx), "ground_truth": x})
            s.n_pipeline_slabs = slabs
            s.shard_update = shard
            assert s._pipelined == force
            ctx.set_option("comm_test_poison_us", poison)
            ctx.profile_reset()
            ctx.profile_enable(True)
            res[(force, slabs, shard, poison)] = s.run_main_iteration(niter=5, positivity=True)
            ctx.profile_enable(False)
            ctx.set_option("comm_test_poison_us", 0)
            n_wait = ctx.profile_get("comm_join_wait")[0]
            n_coll = {k: ctx.profile_get(k)[0] for k in ("allreduce_f32", "reduce_scatter_f32", "allgather_f32")}
            n_fwd = sum(ctx.profile_get(k)[0] for k in ("k_fwd_tile_flat", "k_fwd_tile"))
            if force:
                n_slab = len(s._plan)
                n_fslab = sum(1 for _, _, (f0, f1) in s._plan if f1 > f0)   # a one-column first slab has no forward columns ready yet
                assert n_slab == len(np.unique(np.linspace(0, 6, min(slabs, 6) + 1).astype(int))) - 1 and n_fslab >= n_slab - 1
                assert sorted(c for _, _, (f0, f1) in s._plan for c in range(f0, f1)) == list(range(6))     # every tile column once
                # every collective is waited for exactly once per iteration: one all-reduce per slab, or a reduce-scatter and an all-gather
                assert n_wait == 5 * n_slab * (2 if shard else 1), (n_wait, n_slab)
                assert n_coll == ({"allreduce_f32": 0, "reduce_scatter_f32": 5 * n_slab, "allgather_f32": 5 * n_slab} if shard else
                                  {"allreduce_f32": 5 * n_slab, "reduce_scatter_f32": 0, "allgather_f32": 0}), n_coll
                assert n_fwd == 1 + 4 * n_fslab, (n_fwd, n_fslab)      # iteration 1 whole; 2..5 slab by slab; none made ahead after the last
            else:
                assert n_wait == 0 and n_fwd == 5
        ref = res[(False, 8, True, 0)]
        for key, got in res.items():
            assert rel_max(got[0], ref[0]) < 2e-6 and np.allclose(got[1], ref[1], rtol=1e-6), (tilt, key)
        assert ref[1][-1] < ref[1][0]
    comm.close()


@pytest.mark.timeout(900)
def test_stream_overlap_waits_have_teeth(capsys):
    """the stream-overlap machinery of the sharded solvers (per-collective events, tomo_comm_wait_next /
    _wait_next_gather, the next forward projection behind in-flight all-gathers) checked where it CAN race -- a fresh child process
    whose only context owns its hardware queues (tests/_poison_child.py): (1) the control is an assertion: an un-waited read sees the
    doubled buffer; (2) poisoned pipelined SIRT and CGLS equal the plain sequences; (3) with ONE wait dropped they do not."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_poison_child.py")], capture_output=True, text=True, timeout=800,
                       env=dict(os.environ, NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo")))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[-1])
    try:      # kept for the round's profiles/ (the GPU box merges gpurun_out/ back)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "poison_child_last.json"), "w") as f:
            f.write(lines[-1] + "\n")
    except OSError:
        pass
    with capsys.disabled():
        print("overlap control in a fresh process: un-waited read %.0f (doubled: %.0f), waited %.0f; poisoned solvers worst rel %.1e; "
              "least deviation with one wait dropped %.1e" % (out["unwaited"], 4096 * 36.0, out["waited"], out["poisoned_worst_rel"], out["wait_dropped_least_rel"]))
    assert out["control_ok"], (out["unwaited"], out["waited"])
    assert out["solvers_ok"], [c for c in out["cases"] if c[4] == "poisoned"]
    assert out["mutation_ok"], [c for c in out["cases"] if c[4] != "poisoned"]
    assert r.returncode == 0
    if not out["control_raced"]:
        # the un-waited control read never caught the doubled buffer in four attempts of growing idle time (a wall-clock race): the
        # solver and mutation assertions above are the hard ones -- a dropped wait DID change the results in this very process
        pytest.xfail("overlap control: could not race on this box (%s); solvers and mutation cases passed" % out["control_attempts"])


def test_volume_residency_is_explicit(shepp32):
    """Like the reference, projection_gradient re-reads `rec` on every call; a volume stays resident (upload and zero-padded
    staging skipped) only while the caller has pinned it."""
    from tomography_alignment_amd.utilities import projection_operators
    from tomography_alignment_amd.utilities.generate_phantom import shepp3d
    N = 128
    geo = geom(1, N)
    P = projection_operators.ProjectionMatrix(geo)
    pose = dict(alpha=0.01, beta=-0.02, phi=0.8, xyz_shift=np.array([0.5, 0., -0.7]), cor_shift=np.zeros(3))
    x = shepp3d(N)                                      # z = 0 plane and corner voxels empty (strided fingerprints miss edits)
    assert not x[:, :, 0].any() and x.ravel()[-1] == 0
    p1, g1 = P.projection_gradient(x, **pose)
    assert not P._vol_staged
    x *= 2.0                                            # in-place edit of an unpinned host volume: must be seen
    p2, g2 = P.projection_gradient(x, **pose)
    assert rel_max(p2, 2.0 * p1) < 1e-6 and rel_max(g2, 2.0 * g1) < 1e-6
    x[x < 0.5] = 0.0
    x *= 0.5
    P.pin_volume(x)
    p3, _ = P.projection_gradient(x, **pose)
    p4, _ = P.projection_gradient(x, **pose)            # pinned: staged copy reused
    assert np.array_equal(p3, p4) and P._vol_staged
    x *= 3.0                                            # the caller breaks the promise ...
    P.invalidate_volume()                               # ... and says so
    p5, _ = P.projection_gradient(x, **pose)
    assert rel_max(p5, 3.0 * p3) < 1e-6
    P.unpin_volume()
    d = P.backend.upload(x)
    p6, _ = P.projection_gradient(d, **pose)            # device buffer, unpinned: re-staged on every call
    d.upload(2.0 * x)                                   # mutated in place behind the operator's back
    p7, _ = P.projection_gradient(d, **pose)
    assert rel_max(p6, p5) < 1e-7 and rel_max(p7, 2.0 * p5) < 1e-6


def test_sirt_regularized_gradient_descent_vs_reference_golden_g12(shepp32, capsys):
    """recon/sirt.py::SIRT.run_regularized_gradient_descent (reference :109-180: Tikhonov gradient descent, scipy strong-Wolfe line search on
    my_f / my_fp) on the GPU operator against golden G12 -- the reference's class on its own CSR -- at 1e-5; my_f / my_fp themselves too."""
    from tomography_alignment_amd.recon import sirt as sirt_mod
    g5, g = golden("g5_sirt"), golden("g12_sirt_regularized_gd")
    geo = geom(16, 32)
    angles = np.array([g5["phi"], g5["alpha"], g5["beta"]]).T
    for tag, gt in (("a", None), ("b", shepp32)):
        opts = {} if gt is None else {"ground_truth": gt.copy()}
        s = sirt_mod.SIRT(geo, g5["b"].copy(), angles, g5["xyz"], options=opts)
        rec, err = s.run_regularized_gradient_descent(niter=int(g["nit_" + tag]), reg_param=float(g["reg_" + tag]), positivity=bool(g["pos_" + tag]))
        e, er = rel_max(rec, g["rec_" + tag]), float(np.max(np.abs(err - g["err_" + tag]) / g["err_" + tag]))
        with capsys.disabled():
            print("Tikhonov gradient descent vs the reference's class (G12 %s): rec rel-max %.2e, rms rel %.2e" % (tag, e, er))
        assert len(err) == len(g["err_" + tag]) and e < 1e-5 and er < 1e-5
    assert abs(sirt_mod.my_f(g["f_x"], s.proj_mat, g5["b"], 0.7) / float(g["my_f"]) - 1) < 1e-6
    assert rel_max(sirt_mod.my_fp(g["f_x"], s.proj_mat, g5["b"], 0.7), g["my_fp"]) < 1e-5
