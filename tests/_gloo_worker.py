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
    c = cgls_mpi.CGLS(comm, geo, g["b"].copy(), angles, g["xyz"], options={"_backend": OracleBackend(shard)})
    crec, cerr = c.run_main_iteration(niter=4)
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
    if comm.rank == 0:
        np.savez(out_path, rec=rec, err=err, crec=crec, cerr=cerr, n_allreduce_sirt=n_allreduce_sirt, cor=cor,
                 n_slab_sirt=n_slab_sirt, pipelined=pipelined, n_rs=n_rs, n_ag=n_ag, n_wg=n_wg, slab_sizes=slab_sizes, rec_a=rec_a, err_a=err_a,
                 n_slab_allreduce_form=n_slab_allreduce_form, err_gt_sharded=err_gt_sharded, err_gt_allreduce=err_gt_allreduce, rank_spread=rank_spread, n_fwd_whole=n_fwd_whole, rec_d=rec_d, err_d=err_d,
                 declined_pipelined=declined["pipelined"], declined_n_vol=declined["n_vol"], declined_n_slab=declined["n_slab"],
                 rec_g=rec_g, err_g=err_g, rec_p=rec_p, err_p=err_p, rec_r=rec_r, err_r=err_r,
                 align_x=ares["x"], align_fun=ares["fun"], align_true=true, align_nfev=ares["nfev"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
