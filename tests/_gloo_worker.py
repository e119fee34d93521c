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
    n_allreduce_sirt = comm.n_vol_allreduce
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
                 align_x=ares["x"], align_fun=ares["fun"], align_true=true, align_nfev=ares["nfev"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
