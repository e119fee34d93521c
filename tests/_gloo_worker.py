"""Worker of tests/test_dist_gloo.py: one rank of an angle-sharded SIRT / CGLS run over torch.distributed
(gloo, CPU) with the oracle-backed stand-in backend.  Rank 0 writes the results."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main(out_path):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="env://")
    from backends import OracleBackend, GlooComm
    from tomography_alignment_amd.utilities.geometry import Geometry
    from tomography_alignment_amd.recon import sirt_mpi, cgls_mpi

    comm = GlooComm()
    g = np.load(os.path.join(HERE, "golden", "g5_sirt.npz"))
    N, n_proj = 32, 16
    rng = np.random.default_rng(42)
    cor = np.zeros((n_proj, 3))
    cor[:, 0] = rng.uniform(-1, 1, n_proj)         # per-angle centre-of-rotation shifts must follow their angles
    geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2), cor_shift=cor)
    angles = np.array([g["phi"], g["alpha"], g["beta"]]).T
    my = np.array_split(np.arange(n_proj), comm.size)[comm.rank]
    shard = sirt_mpi.SIRT._shard_geometry(geo, my)
    s = sirt_mpi.SIRT(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard)})
    assert np.array_equal(s.my_index, my) and s.proj_mat.shape[0] == my.size * N * N
    rec, err = s.run_main_iteration(niter=6, positivity=True)
    n_allreduce_sirt, n_slab_sirt, pipelined = comm.n_vol_allreduce, comm.n_slab_allreduce, s._pipelined
    n_rs, n_ag, n_wg = getattr(comm, "n_reduce_scatter", 0), getattr(comm, "n_allgather", 0), getattr(comm, "n_wait_gather", 0)
    slab_sizes = np.array(comm.slab_sizes[:max(1, n_rs + n_slab_sirt) // max(1, len(err))], np.int64)     # the first iteration's collectives
    n_fwd_whole = s.be.calls["forward"]
    # the round-3 form of the pipelined iteration (all-reduce per slab, the identical update on every rank) must stay equivalent
    s0_ = comm.n_slab_allreduce
    sa = sirt_mpi.SIRT(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard)})
    sa.shard_update = False
    rec_a, err_a = sa.run_main_iteration(niter=6, positivity=True)
    n_slab_allreduce_form = comm.n_slab_allreduce - s0_
    # ... with a ground truth too: EVERY rank must see the same error curve (the stop rule hangs on it: a rank that stopped alone would
    # leave its peers in a collective), whichever form sums the slabs
    curves = []
    for shard_update in (True, False):
        sgt = sirt_mpi.SIRT(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard), "ground_truth": rec})
        sgt.shard_update = shard_update
        _, e_gt = sgt.run_main_iteration(niter=3, positivity=True)
        curves.append((e_gt, comm.allreduce_max(float(e_gt[-1])) - (-comm.allreduce_max(-float(e_gt[-1])))))
    err_gt_sharded, err_gt_allreduce = curves[0][0], curves[1][0]
    rank_spread = max(curves[0][1], curves[1][1])          # largest difference between the ranks' last rms value (0 when they agree)
    # the same run with a stand-in backend that DECLINES the tile kernels on the last rank only (an angle block holding a pose
    # tilted beyond their domain): the decision is collective, so every rank must take the plain sequence -- one whole-volume
    # all-reduce per iteration on every rank, no slab all-reduce anywhere -- and the result must not change
    v0, s0 = comm.n_vol_allreduce, comm.n_slab_allreduce
    sd = sirt_mpi.SIRT(comm, geo, g["b"].copy(), angles, g["xyz"],
                       options={"_backend": OracleBackend(shard, declines_tiles=(comm.rank == comm.size - 1))})
    rec_d, err_d = sd.run_main_iteration(niter=6, positivity=True)
    declined = dict(pipelined=sd._pipelined, n_vol=comm.n_vol_allreduce - v0, n_slab=comm.n_slab_allreduce - s0)
    # ... and with a ground truth (the error sum accumulates over the slabs on the "device") and a forced pipeline at world 1
    comm.force_pipeline = True
    sg = sirt_mpi.SIRT(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard), "ground_truth": g["gt"] if "gt" in g else rec})
    sg.n_pipeline_slabs = 2
    rec_g, err_g = sg.run_main_iteration(niter=4)
    comm.force_pipeline = False
    sp = sirt_mpi.SIRT(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard, declines_tiles=True),
                                                                          "ground_truth": g["gt"] if "gt" in g else rec})
    rec_p, err_p = sp.run_main_iteration(niter=4)
    assert sg._pipelined and not sp._pipelined
    # Tikhonov gradient descent (recon/sirt_mpi.py:148-: the data terms of f, f' and the gradient summed over the ranks)
    rr = sirt_mpi.SIRT(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard)})
    rec_r, err_r = rr.run_regularized_gradient_descent(niter=3, reg_param=0.5, positivity=True)
    # CGLS: pipelined at world > 1 (reduce-scatter per slab, gamma accumulated over the own pieces, p updated piecewise and all-gathered,
    # the next A p projected slab by slab behind the all-gathers), plain at world 1; forced pipeline / all-reduce form / plain must agree
    rs0, ag0, sl0, v0c = getattr(comm, "n_reduce_scatter", 0), getattr(comm, "n_allgather", 0), comm.n_slab_allreduce, comm.n_vol_allreduce
    c = cgls_mpi.CGLS(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard)})
    crec, cerr = c.run_main_iteration(niter=4)
    c_counts = np.array([getattr(comm, "n_reduce_scatter", 0) - rs0, getattr(comm, "n_allgather", 0) - ag0, comm.n_slab_allreduce - sl0, comm.n_vol_allreduce - v0c,
                         int(c._pipelined), c.be.calls["forward"]])
    cvar = {}
    for tag, force, shard_upd, slabs, gt_ in (("forced", True, True, 8, None), ("allreduce", True, False, 8, None), ("plain", False, True, 1, None), ("gt", True, True, 2, rec)):
        comm.force_pipeline = force
        o_ = {"_backend": OracleBackend(shard)}
        if gt_ is not None:
            o_["ground_truth"] = gt_
        cc = cgls_mpi.CGLS(comm, geo, g["b"].copy(), angles, g["xyz"], options=o_)
        cc.shard_update, cc.n_pipeline_slabs = shard_upd, slabs
        cvar[tag] = cc.run_main_iteration(niter=4)
        comm.force_pipeline = False
    # sharded alignment: projections split over the ranks, replicated volume, one table all-reduce at the end
    from oracle import oracle as orc
    from tomography_alignment_amd import alignment
    Na, na = 16, 4
    xa = orc.shepp3d(Na).astype(np.float32)
    phia = np.array([0.4, 1.1, 1.9, 2.6])
    true = np.column_stack([[0.8, -0.5, 0.3, -0.9], [-0.4, 0.7, -0.6, 0.2], np.deg2rad([0.5, -0.4, 0.3, -0.2]), np.deg2rad([-0.3, 0.2, 0.4, -0.5])])
    oga = orc.Geo(1, np.array([Na] * 3), np.ones(3), np.array([Na, Na]), np.ones(2))
    ba = np.array([orc.projection_gradient(oga, xa, true[i, 2], true[i, 3], phia[i], np.array([true[i, 0], 0., true[i, 1]]), np.zeros(3))[0]
                   for i in range(na)])
    geoa = Geometry(na, np.array([Na] * 3), np.ones(3), np.array([Na, Na]), np.ones(2))
    bounds = ((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02))
    ares = alignment.align_projections_sharded(comm, OracleBackend(geoa), xa, ba, phia, letters="xzab", bounds=bounds)
    # config 5 end to end on N ranks: the outer loop of examples/align_rigid.py:27-59 -- angle-sharded SIRT, then
    # every rank aligns its own projections against the replicated reconstruction, the pose table summed once per outer iteration
    from tomography_alignment_amd.examples import align_rigid
    from tomography_alignment_amd.recon import sirt_mpi as _sm
    ne = 6
    phie = np.linspace(0.2, 2.9, ne)
    rng_e = np.random.default_rng(7)
    te = np.column_stack([rng_e.uniform(-1.5, 1.5, ne), rng_e.uniform(-1.5, 1.5, ne), np.deg2rad(rng_e.uniform(-0.8, 0.8, ne)), np.deg2rad(rng_e.uniform(-0.8, 0.8, ne))])
    oge = orc.Geo(ne, np.array([Na] * 3), np.ones(3), np.array([Na, Na]), np.ones(2))
    xyze = np.zeros((ne, 3))
    xyze[:, 0], xyze[:, 2] = te[:, 0], te[:, 1]
    be_ = orc.forward(oge, xa, alpha=te[:, 2], beta=te[:, 3], phi=phie, xyz_shift=xyze).astype(np.float32).reshape(ne, Na, Na)
    datae = dict(projections=be_, phi=phie, phantom=xa, xyz=xyze, alpha=te[:, 2], beta=te[:, 3])
    geoe = Geometry(ne, np.array([Na] * 3), np.ones(3), np.array([Na, Na]), np.ones(2))
    mine_e = np.array_split(np.arange(ne), comm.size)[comm.rank]
    obe = OracleBackend(_sm.SIRT._shard_geometry(geoe, mine_e))
    tight = {"options": {"ftol": 1e-15, "gtol": 1e-11}}
    e_rec, e_a, e_b, e_xyz, e_hist = align_rigid.run(datae, n_outer=2, sirt_iters=5, verbose=False, backend=obe, comm=comm, align_kwargs=tight)
    e_uploaded_rows = obe.n_uploaded // (Na * Na)      # measured rows this rank put on the "device" (+ the Na ground-truth planes, uploaded once)
    e_spread = max(comm.allreduce_max(float(v)) + comm.allreduce_max(-float(v)) for v in np.concatenate([e_a, e_b, e_xyz.ravel()]))
    # ... and HALF BY HALF against an unsharded loop run beside it on this very rank, both halves fed the same inputs.  (The composition is
    # not comparable at float32 accuracy: L-BFGS-B on the piecewise-trilinear cost amplifies a 1e-7 perturbation of the reconstruction
    # to 0.1 px -- the unsharded loop against ITSELF with such a perturbation, /tmp probe of round 5 -- so each half is compared on
    # identical inputs: sharded SIRT == unsharded SIRT at 1e-5; sharded pass == unsharded pass exactly.)
    from tomography_alignment_amd.comm import SingleComm
    ref = align_rigid.OuterLoop(datae, backend=OracleBackend(geoe), comm=SingleComm())       # sirt_mpi's rules (guard 1e-8, stop k > 1), one rank
    shd = align_rigid.OuterLoop(datae, backend=OracleBackend(_sm.SIRT._shard_geometry(geoe, mine_e)), comm=comm)
    st = {}
    for stage in (0, 1):
        (k_r, err_r), (k_s, err_s) = ref.reconstruct(5), shd.reconstruct(5)
        a_r, a_s = ref.download(), shd.download()
        st["sirt%d_rec" % stage] = float(np.max(np.abs(a_s - a_r)) / np.max(np.abs(a_r)))
        st["sirt%d_err" % stage] = float(np.max(np.abs(err_s - err_r) / err_r)) if k_r == k_s else 1.0
        shd.d_rec.upload(a_r)                                  # the same reconstruction into both alignment passes ...
        r_r, r_s = ref.align(**tight), shd.align(**tight)
        st["align%d_x" % stage] = float(np.max(np.abs(r_s["x"] - r_r["x"])))
        st["align%d_fun" % stage] = float(np.max(np.abs(r_s["fun"] - r_r["fun"])))
        st["align%d_nfev" % stage] = int(np.max(np.abs(r_s["nfev"] - r_r["nfev"])))
        shd.alpha_rec, shd.beta_rec, shd.xyz_rec = ref.alpha_rec.copy(), ref.beta_rec.copy(), ref.xyz_rec.copy()     # ... and the same poses into the next SIRT
    st["pose_moved"] = float(np.abs(ref.xyz_rec).max())
    if comm.rank == 0:
        extra = {"c_%s_%s" % (t, n): v[i] for t, v in cvar.items() for i, n in enumerate(("rec", "err"))}
        extra.update({"st_" + k: v for k, v in st.items()})
        np.savez(out_path, c_counts=c_counts, rec=rec, e_rec=e_rec, e_a=e_a, e_b=e_b, e_xyz=e_xyz, e_rmse=np.array([h["rmse"] for h in e_hist]),
                 e_residual=np.array([h["residual"] for h in e_hist]), e_shift_err=np.array([h["shift_err_px"] for h in e_hist]),
                 e_uploaded_rows=e_uploaded_rows, e_true=te, e_spread=e_spread, **extra, err=err, crec=crec, cerr=cerr, n_allreduce_sirt=n_allreduce_sirt, cor=cor,
                 n_slab_sirt=n_slab_sirt, pipelined=pipelined, n_rs=n_rs, n_ag=n_ag, n_wg=n_wg, slab_sizes=slab_sizes, rec_a=rec_a, err_a=err_a,
                 n_slab_allreduce_form=n_slab_allreduce_form, err_gt_sharded=err_gt_sharded, err_gt_allreduce=err_gt_allreduce, rank_spread=rank_spread, n_fwd_whole=n_fwd_whole, rec_d=rec_d, err_d=err_d,
                 declined_pipelined=declined["pipelined"], declined_n_vol=declined["n_vol"], declined_n_slab=declined["n_slab"],
                 rec_g=rec_g, err_g=err_g, rec_p=rec_p, err_p=err_p, rec_r=rec_r, err_r=err_r,
                 align_x=ares["x"], align_fun=ares["fun"], align_true=true, align_nfev=ares["nfev"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
