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
    if comm.rank == 0:
        np.savez(out_path, rec=rec, err=err, crec=crec, cerr=cerr, n_allreduce_sirt=n_allreduce_sirt, cor=cor)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
