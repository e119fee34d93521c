"""Worker of tests/test_dist_gloo.py::test_world_8_*: ONE rank of the 8-rank shape BASELINE configs 4 / 5 name -- angle-sharded SIRT, CGLS and one
outer iteration of examples/align_rigid.run(comm=) -- over torch.distributed (gloo, CPU) with the oracle-backed stand-in backend, at a size
8 vCPUs finish in seconds.  The same program runs at world 1 as the reference.  What it pins before the first real 8-GPU run
(VERDICT r5 next 6): a slab plan of 8 slabs cut into 8 pieces each with left-over voxels on some slabs, angle blocks of unequal size
(20 angles: 3 3 3 3 2 2 2 2; 10 projections: 2 2 1 1 1 1 1 1), collective counts identical on every rank, one pose table on every rank.
Rank 0 writes the results."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def counts(comm):
    return np.array([comm.n_vol_allreduce, comm.n_slab_allreduce, getattr(comm, "n_reduce_scatter", 0), getattr(comm, "n_allgather", 0),
                     comm.n_wait, getattr(comm, "n_wait_gather", 0)], np.int64)


def main(out_path):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="env://")
    from backends import OracleBackend, GlooComm
    from oracle import oracle as orc
    from tomography_alignment_amd.comm import SingleComm
    from tomography_alignment_amd.examples import align_rigid
    from tomography_alignment_amd.recon import sirt_mpi, cgls_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry

    comm = GlooComm()
    comm.force_pipeline = True                        # world 1 takes the slab pipeline too: same code path as the 8 ranks
    out = {}
    # ---- SIRT / CGLS: 123 x 15 x 17 voxels = 8 tile columns of 16 -> 8 slabs of 15, 16 x 6, 12 planes of 255 voxels: 3825 = 8 x 478 + 1,
    # 4080 = 8 x 510, 3060 = 8 x 382 + 4 -- pieces by reduce-scatter / all-gather, the 1 and 4 left-over voxels by a small all-reduce
    shape, ndet, n_proj = (123, 15, 17), (123, 17), 20
    rng = np.random.default_rng(8)
    x = np.zeros(shape, np.float32)
    x[20:100, 3:12, 2:15] = rng.uniform(0.2, 1.0, (80, 9, 13)).astype(np.float32)
    phi = np.linspace(0.05, np.pi - 0.05, n_proj)
    cor = np.zeros((n_proj, 3))
    cor[:, 0] = rng.uniform(-1, 1, n_proj)
    xyz = np.zeros((n_proj, 3))
    xyz[:, 0], xyz[:, 2] = rng.uniform(-2, 2, n_proj), rng.uniform(-2, 2, n_proj)
    geo = Geometry(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor)
    og = orc.Geo(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor)
    b = orc.forward(og, x, phi=phi, xyz_shift=xyz).astype(np.float32).reshape(n_proj, -1)
    ang = np.array([phi, 0 * phi, 0 * phi]).T
    mine = np.array_split(np.arange(n_proj), comm.size)[comm.rank]
    out["block_sizes"] = np.array([np.array_split(np.arange(n_proj), comm.size)[r].size for r in range(comm.size)])
    shard = sirt_mpi.SIRT._shard_geometry(geo, mine)
    c0 = counts(comm)
    s = sirt_mpi.SIRT(comm, geo, b.copy(), ang, xyz, options={"_backend": OracleBackend(shard), "ground_truth": x})
    assert s.n_pipeline_slabs == 8
    c1 = counts(comm)
    rec, err = s.run_main_iteration(niter=3, positivity=True)
    c2 = counts(comm)
    out.update(sirt_rec=rec, sirt_err=err, sirt_pipelined=bool(s._iter_pipelined), sirt_init_counts=c1 - c0, sirt_counts=c2 - c1,
               plan_x=np.array([[lo, hi] for _, (lo, hi), _ in s._plan]), slab_sizes=np.array(comm.slab_sizes[:16], np.int64))
    c = cgls_mpi.CGLS(comm, geo, b.copy(), ang, xyz, options={"_backend": OracleBackend(shard)})
    c3 = counts(comm)
    crec, cerr = c.run_main_iteration(niter=3)
    c4 = counts(comm)
    out.update(cgls_rec=crec, cgls_err=cerr, cgls_pipelined=bool(getattr(c, "_iter_pipelined", False)), cgls_counts=c4 - c3)
    # ---- one outer iteration of examples/align_rigid.run on these ranks: 16^3, 10 projections (blocks 2 2 1 1 1 1 1 1; one tile column -> the
    # plain sequence of collectives: a whole-volume all-reduce per SIRT iteration), then each half against the unsharded loop on the same inputs
    Na, ne = 16, 10
    xa = orc.shepp3d(Na).astype(np.float32)
    phie = np.linspace(0.2, 2.9, ne)
    re_ = np.random.default_rng(7)
    te = np.column_stack([re_.uniform(-1.5, 1.5, ne), re_.uniform(-1.5, 1.5, ne), np.deg2rad(re_.uniform(-0.8, 0.8, ne)), np.deg2rad(re_.uniform(-0.8, 0.8, ne))])
    oge = orc.Geo(ne, np.array([Na] * 3), np.ones(3), np.array([Na, Na]), np.ones(2))
    xyze = np.zeros((ne, 3))
    xyze[:, 0], xyze[:, 2] = te[:, 0], te[:, 1]
    be_ = orc.forward(oge, xa, alpha=te[:, 2], beta=te[:, 3], phi=phie, xyz_shift=xyze).astype(np.float32).reshape(ne, Na, Na)
    datae = dict(projections=be_, phi=phie, phantom=xa, xyz=xyze, alpha=te[:, 2], beta=te[:, 3])
    geoe = Geometry(ne, np.array([Na] * 3), np.ones(3), np.array([Na, Na]), np.ones(2))
    mine_e = np.array_split(np.arange(ne), comm.size)[comm.rank]
    tight = {"options": {"ftol": 1e-15, "gtol": 1e-11}}
    c5 = counts(comm)
    obe = OracleBackend(sirt_mpi.SIRT._shard_geometry(geoe, mine_e))
    e_rec, e_a, e_b, e_xyz, e_hist = align_rigid.run(datae, n_outer=1, sirt_iters=4, verbose=False, backend=obe, comm=comm, align_kwargs=tight)
    c6 = counts(comm)
    out.update(e_counts=c6 - c5, e_a=e_a, e_b=e_b, e_xyz=e_xyz, e_rmse=np.array([h["rmse"] for h in e_hist]), e_shift_err=np.array([h["shift_err_px"] for h in e_hist]),
               e_uploaded_rows=obe.n_uploaded // (Na * Na), e_true=te)
    out["e_spread"] = max(comm.allreduce_max(float(v)) + comm.allreduce_max(-float(v)) for v in np.concatenate([e_a, e_b, e_xyz.ravel()]))
    ref = align_rigid.OuterLoop(datae, backend=OracleBackend(geoe), comm=SingleComm())
    shd = align_rigid.OuterLoop(datae, backend=OracleBackend(sirt_mpi.SIRT._shard_geometry(geoe, mine_e)), comm=comm)
    (k_r, err_r), (k_s, err_s) = ref.reconstruct(4), shd.reconstruct(4)
    a_r, a_s = ref.download(), shd.download()
    out["st_sirt_rec"] = float(np.max(np.abs(a_s - a_r)) / np.max(np.abs(a_r)))
    out["st_sirt_err"] = float(np.max(np.abs(err_s - err_r) / err_r)) if k_r == k_s else 1.0
    shd.d_rec.upload(a_r)
    r_r, r_s = ref.align(**tight), shd.align(**tight)
    out["st_align_x"] = float(np.max(np.abs(r_s["x"] - r_r["x"])))
    out["st_align_nfev"] = int(np.max(np.abs(r_s["nfev"] - r_r["nfev"])))
    # ---- more ranks than angles: 6 angles on 8 ranks -- ranks 6 and 7 own NO projection (np.array_split gives them empty blocks), yet must issue every
    # collective of every iteration (their partial volumes are zero) and end with the same reconstruction; SIRT pipelined, CGLS, and an alignment pass
    # in which two ranks have nothing to align (align_projections_sharded's empty-block branch)
    n6 = 6
    phi6 = np.linspace(0.1, 3.0, n6)
    geo6 = Geometry(n6, np.array(shape), np.ones(3), np.array(ndet), np.ones(2))
    og6 = orc.Geo(n6, np.array(shape), np.ones(3), np.array(ndet), np.ones(2))
    b6 = orc.forward(og6, x, phi=phi6).astype(np.float32).reshape(n6, -1)
    ang6 = np.array([phi6, 0 * phi6, 0 * phi6]).T
    mine6 = np.array_split(np.arange(n6), comm.size)[comm.rank]
    sh6 = sirt_mpi.SIRT._shard_geometry(geo6, mine6)
    c7 = counts(comm)
    s6 = sirt_mpi.SIRT(comm, geo6, b6.copy(), ang6, np.zeros((n6, 3)), options={"_backend": OracleBackend(sh6), "ground_truth": x})
    rec6, err6 = s6.run_main_iteration(niter=2, positivity=True)
    cg6 = cgls_mpi.CGLS(comm, geo6, b6.copy(), ang6, np.zeros((n6, 3)), options={"_backend": OracleBackend(sh6)})
    crec6, cerr6 = cg6.run_main_iteration(niter=2)
    c8 = counts(comm)
    out.update(few_sirt_rec=rec6, few_sirt_err=err6, few_cgls_rec=crec6, few_cgls_err=cerr6, few_counts=c8 - c7, few_my_n=np.array(mine6.size),
               few_pipelined=bool(s6._iter_pipelined))
    out["few_empty_ranks"] = np.array(int(round(comm.allreduce_scalar(1.0 if mine6.size == 0 else 0.0))))
    from tomography_alignment_amd import alignment
    phia = phie[:n6]
    resa = alignment.align_projections_sharded(comm, OracleBackend(Geometry(n6, np.array([Na] * 3), np.ones(3), np.array([Na, Na]), np.ones(2))), xa, be_[:n6], phia,
                                               letters="xzab", bounds=((-3., 3.), (-3., 3.), (-0.02, 0.02), (-0.02, 0.02)), **tight)
    out.update(few_align_x=resa["x"], few_align_nfev=resa["nfev"])
    # ---- rank-uniform bookkeeping: every rank issued the same number of every kind of collective (a mismatch would have hung gloo; this
    # also shows it in the record), taken over the whole program
    total = counts(comm)
    out["counts_spread"] = np.array([comm.allreduce_max(float(v)) + comm.allreduce_max(-float(v)) for v in total])
    if comm.rank == 0:
        np.savez(out_path, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
