"""Worker of tests/test_gpu_dist.py: one rank of an angle-sharded SIRT run with the REAL HIP backend (every rank opens its own
context on GPU 0) and a host-staged torch.distributed/gloo communicator standing in for RCCL (two RCCL ranks cannot share one GPU;
the 1-GPU boxes of this pool cannot run RCCL with more than one rank).  What this exercises that nothing else does: the sharded
solver's slab pipeline (tomo_adjoint_xslab / tomo_forward_xslab / tomo_vec_update_acc on views) with world size 2, i.e. partial
volumes that really differ between ranks and are summed by a collective.  Rank 0 writes the results."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


class HostStagedComm(object):
    """RcclComm's method set over gloo: device buffers are all-reduced through the host.  The asynchronous forms complete at once."""

    def __init__(self, ctx):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.ctx = ctx
        self.n_vol_allreduce = self.n_slab_allreduce = self.n_wait = 0

    def _ar(self, buf):
        import torch
        if buf.size:
            h = buf.download()
            t = torch.from_numpy(h)
            self.dist.all_reduce(t)
            buf.upload(h)
        return buf

    def allreduce_sum_(self, buf):
        self.n_vol_allreduce += 1
        return self._ar(buf)

    def allreduce_sum_async(self, buf):
        self.n_slab_allreduce += 1
        return self._ar(buf)

    def wait_next(self):
        self.n_wait += 1

    # round 4: reduce-scatter (the pieces this rank does not own are left as NaN, as undefined as RCCL leaves them) and all-gather
    def reduce_scatter_sum_async(self, buf, n_per_rank):
        import torch
        n = int(n_per_rank)
        if n:
            seg = buf.view(0, n * self.size)
            h = seg.download()
            self.dist.all_reduce(torch.from_numpy(h))
            mine = h[self.rank * n:(self.rank + 1) * n].copy()
            if self.size > 1:
                h[:] = np.nan
            h[self.rank * n:(self.rank + 1) * n] = mine
            seg.upload(h)
        self.n_rs = getattr(self, "n_rs", 0) + 1
        return buf

    def allgather_async(self, buf, n_per_rank):
        import torch
        n = int(n_per_rank)
        if n:
            seg = buf.view(0, n * self.size)
            h = seg.download()
            parts = [torch.empty(n, dtype=torch.float32) for _ in range(self.size)]
            self.dist.all_gather(parts, torch.from_numpy(h[self.rank * n:(self.rank + 1) * n].copy()))
            for q, part in enumerate(parts):
                h[q * n:(q + 1) * n] = part.numpy()
            seg.upload(h)
        self.n_ag = getattr(self, "n_ag", 0) + 1
        return buf

    def wait_next_gather(self):
        pass

    def join(self):
        pass

    def allreduce_scalar(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t)
        return float(t[0])

    def allreduce_max(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def allreduce_array(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, np.float64))
        self.dist.all_reduce(t)
        a[...] = t.numpy()
        return a

    def barrier(self):
        self.dist.barrier()


def align_rigid_stages(ctx, comm, out):
    """examples/align_rigid's outer loop on this world: the composed run, and each half against an unsharded loop on the
    same context fed the same inputs (see tests/_gloo_worker.py for why the halves and not the composition are compared tightly)."""
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.comm import SingleComm
    from tomography_alignment_amd.examples import align_rigid
    from tomography_alignment_amd.recon import sirt_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry
    from tomography_alignment_amd.utilities.generate_phantom import shepp3d
    N, n_proj = 48, 12
    rng = np.random.default_rng(17)
    x = shepp3d(N).astype(np.float32)
    phi = np.linspace(0.0, np.pi, n_proj)
    alpha, beta = np.deg2rad(rng.uniform(-0.8, 0.8, n_proj)), np.deg2rad(rng.uniform(-0.8, 0.8, n_proj))
    xyz = np.zeros((n_proj, 3))
    xyz[:, 0], xyz[:, 2] = rng.uniform(-1.5, 1.5, n_proj), rng.uniform(-1.5, 1.5, n_proj)
    geo = Geometry(n_proj, np.array([N] * 3), np.ones(3), np.array([N, N]), np.ones(2))
    full = HipBackend(geo, ctx=ctx)
    b = full.forward(_lib.poses_array(phi, alpha, beta, xyz, np.zeros(3)), full.upload(x), full.empty(n_proj * N * N)).download().reshape(n_proj, N, N)

    def from_rank0(a):
        """rank 0's copy on every rank: whatever a rank computes with the forward projectors (float32 atomics into the sinogram: equal up to
        the order of the additions) differs in the last bits from the same computation on another rank -- the measured projections, and the
        unsharded reference reconstruction.  Every rank must start from, compare against and continue from the SAME arrays."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        comm.dist.broadcast(t, src=0)
        return t.numpy()

    b = from_rank0(b)
    data = dict(projections=b, phi=phi, phantom=x, xyz=xyz, alpha=alpha, beta=beta)
    mine = np.array_split(np.arange(n_proj), comm.size)[comm.rank]
    shard_be = lambda: HipBackend(sirt_mpi.SIRT._shard_geometry(geo, mine), ctx=ctx)      # noqa: E731
    comm.force_pipeline = True            # the slab pipeline (reduce-scatter / own piece / all-gather) also at world 1
    rec, a, bb, t, hist, last_loop = align_rigid.run(data, n_outer=2, sirt_iters=8, verbose=False, backend=shard_be(), comm=comm, return_loop=True)
    out["e_rec"], out["e_a"], out["e_b"], out["e_xyz"] = rec, a, bb, t
    out["e_rmse"], out["e_shift_err"] = np.array([h["rmse"] for h in hist]), np.array([h["shift_err_px"] for h in hist])
    out["e_injected"] = np.abs(xyz[:, [0, 2]]).mean()
    out["e_spread"] = max(comm.allreduce_max(float(v)) + comm.allreduce_max(-float(v)) for v in np.concatenate([a, bb, t.ravel()]))
    out["e_pipelined"] = bool(last_loop.solver._iter_pipelined)
    ref = align_rigid.OuterLoop(data, backend=HipBackend(geo, ctx=ctx), comm=SingleComm())
    shd = align_rigid.OuterLoop(data, backend=shard_be(), comm=comm)
    for stage in (0, 1):
        (k_r, err_r), (k_s, err_s) = ref.reconstruct(8), shd.reconstruct(8)
        a_r, a_s = from_rank0(ref.download()), shd.download()
        ref.d_rec.upload(a_r)
        out["st_sirt%d_rec" % stage] = float(np.max(np.abs(a_s - a_r)) / np.max(np.abs(a_r)))
        out["st_sirt%d_err" % stage] = float(np.max(np.abs(err_s - err_r) / err_r)) if k_r == k_s else 1.0
        shd.d_rec.upload(a_r)
        keep = (ref.alpha_rec.copy(), ref.beta_rec.copy(), ref.xyz_rec.copy())
        r_0 = ref.align()                                      # the same pass twice from the same state: bit-identical (round 6)
        ref.alpha_rec, ref.beta_rec, ref.xyz_rec = keep
        r_r, r_s = ref.align(), shd.align()
        out["st_repeat%d_x" % stage] = float(np.max(np.abs(r_0["x"] - r_r["x"])))
        r_r["x"], r_r["fun"] = from_rank0(r_r["x"]), from_rank0(r_r["fun"])
        ref.alpha_rec, ref.beta_rec, ref.xyz_rec = from_rank0(ref.alpha_rec), from_rank0(ref.beta_rec), from_rank0(ref.xyz_rec)
        # Round 6: the fused reduction is deterministic (fixed-order second stage), so the pass reproduces itself and the sharded pass is the
        # unsharded one bit for bit; the EVALUATIONS the pass is made of are compared as well -- the same poses (the unsharded pass's result)
        # through both evaluators: the rank's own-row table, its row map and its centre-of-rotation shifts against the full table
        out["st_align%d_x" % stage] = float(np.max(np.abs(r_s["x"] - r_r["x"])))
        out["st_align%d_fun" % stage] = float(np.max(np.abs(r_s["fun"] - r_r["fun"]) / np.maximum(np.abs(r_r["fun"]), 1e-30)))
        from tomography_alignment_amd import alignment
        poses6 = np.zeros((n_proj, 6))
        poses6[:, 0], poses6[:, 1], poses6[:, 2] = phi, ref.alpha_rec, ref.beta_rec
        poses6[:, 3:6] = ref.xyz_rec
        ev_r = alignment.BatchEvaluator(ref.be, ref.d_rec, ref.d_b, indices=np.arange(n_proj), n_all=n_proj)
        ev_s = alignment.BatchEvaluator(shd.be, shd.d_rec, shd.d_b, indices=mine, n_all=n_proj)
        c_r, g_r = ev_r.evaluate(mine, poses6[mine])
        ev_r.close()
        c_s, g_s = ev_s.evaluate(mine, poses6[mine])
        ev_s.close()
        out["st_eval%d_cost" % stage] = float(np.max(np.abs(c_s - c_r) / np.abs(c_r)))
        out["st_eval%d_grad" % stage] = float(np.max(np.abs(g_s - g_r)) / np.max(np.abs(g_r)))
        shd.alpha_rec, shd.beta_rec, shd.xyz_rec = ref.alpha_rec.copy(), ref.beta_rec.copy(), ref.xyz_rec.copy()
    out["st_pose_moved"] = float(np.abs(ref.xyz_rec).max())
    comm.force_pipeline = False


def main(out_path):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="env://")
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.recon import sirt_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry

    ctx = _lib.Context(0)
    comm = HostStagedComm(ctx)
    # 6 tile columns, ragged in every axis; 200 planes = two 128-plane blocks of the flat forward / four 64-plane chunks of the gather
    # back-projection, the object in planes 70 .. 149 only: all-zero images, empty sinogram planes and z chunks that end at once take part
    shape, ndet, n_proj = (80, 40, 200), (72, 210), 12
    rng = np.random.default_rng(11)
    x = np.zeros(shape, np.float32)
    x[10:70, 6:34, 70:150] = rng.uniform(0.2, 1.0, (60, 28, 80)).astype(np.float32)
    phi = np.linspace(0.05, np.pi - 0.05, n_proj)
    cor = np.zeros((n_proj, 3))
    cor[:, 0] = rng.uniform(-1, 1, n_proj)                       # per-angle COR shifts must follow their angles into the shards
    geo = Geometry(n_proj, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor)
    out = {}
    for tag, tilt in (("flat", 0.0), ("tilted", 1.0)):
        alpha, beta = np.deg2rad(tilt * rng.uniform(-1.5, 1.5, n_proj)), np.deg2rad(tilt * rng.uniform(-1.5, 1.5, n_proj))
        xyz = np.zeros((n_proj, 3))
        xyz[:, 0], xyz[:, 2] = rng.uniform(-2, 2, n_proj), rng.uniform(-2, 2, n_proj)
        ang = np.array([phi, alpha, beta]).T
        # measured projections: every rank computes the whole set with an unsharded operator (same on all ranks)
        full = HipBackend(geo, ctx=ctx)
        b = full.forward(_lib.poses_array(phi, alpha, beta, xyz, cor), full.upload(x), full.empty(n_proj * ndet[0] * ndet[1])).download().reshape(n_proj, -1)
        mine = np.array_split(np.arange(n_proj), comm.size)[comm.rank]
        for mode in ("pipelined", "allreduce", "plain"):     # reduce-scatter / all-gather slabs; all-reduce slabs (round 3); one whole-volume all-reduce
            comm.force_pipeline = mode != "plain"
            s = sirt_mpi.SIRT(comm, geo, b.copy(), ang, xyz, options={"_backend": HipBackend(sirt_mpi.SIRT._shard_geometry(geo, mine), ctx=ctx), "ground_truth": x})
            if mode == "plain":
                s.n_pipeline_slabs = 1
            s.shard_update = mode == "pipelined"
            v0, s0, r0, a0 = comm.n_vol_allreduce, comm.n_slab_allreduce, getattr(comm, "n_rs", 0), getattr(comm, "n_ag", 0)
            rec, err = s.run_main_iteration(niter=5, positivity=True)
            out["%s_%s_rec" % (tag, mode)], out["%s_%s_err" % (tag, mode)] = rec, err
            out["%s_%s_nvol" % (tag, mode)], out["%s_%s_nslab" % (tag, mode)] = comm.n_vol_allreduce - v0, comm.n_slab_allreduce - s0
            out["%s_%s_nrs" % (tag, mode)], out["%s_%s_nag" % (tag, mode)] = getattr(comm, "n_rs", 0) - r0, getattr(comm, "n_ag", 0) - a0
            out["%s_%s_pipelined" % (tag, mode)] = s._iter_pipelined
        # CGLS the same three ways (round 5: reduce-scatter per slab, gamma accumulated on the device over the own pieces, p updated piecewise and
        # all-gathered with the next A p behind it; scalars through device accumulators + the communicator's small all-reduce)
        from tomography_alignment_amd.recon import cgls_mpi
        for mode in ("pipelined", "allreduce", "plain"):
            comm.force_pipeline = mode != "plain"
            c = cgls_mpi.CGLS(comm, geo, b.copy(), ang, xyz, options={"_backend": HipBackend(sirt_mpi.SIRT._shard_geometry(geo, mine), ctx=ctx)})
            if mode == "plain":
                c.n_pipeline_slabs = 1
            c.shard_update = mode == "pipelined"
            r0, a0, s0 = getattr(comm, "n_rs", 0), getattr(comm, "n_ag", 0), comm.n_slab_allreduce
            crec, cerr = c.run_main_iteration(niter=5)
            out["%s_cgls_%s_rec" % (tag, mode)], out["%s_cgls_%s_err" % (tag, mode)] = crec, cerr
            out["%s_cgls_%s_counts" % (tag, mode)] = np.array([getattr(comm, "n_rs", 0) - r0, getattr(comm, "n_ag", 0) - a0, comm.n_slab_allreduce - s0,
                                                               int(getattr(c, "_iter_pipelined", False))])
    align_rigid_stages(ctx, comm, out)
    # ---- a rank without angles on the REAL backend: one projection on this world (at world 2 rank 1 owns nothing: zero-row tables, no projector
    # call, every collective issued), plain and pipelined; the one-rank answer must come out
    x1 = np.zeros((32, 32, 32), np.float32)
    x1[8:24, 10:22, 6:26] = rng.uniform(0.2, 1.0, (16, 12, 20)).astype(np.float32)
    geo1 = Geometry(1, np.array([32] * 3), np.ones(3), np.array([32, 32]), np.ones(2))
    ang1 = np.array([[0.7, 0.0, 0.0]])
    b1 = HipBackend(geo1, ctx=ctx)
    p1 = b1.forward(_lib.poses_array([0.7], [0.0], [0.0], np.zeros((1, 3)), np.zeros(3)), b1.upload(x1), b1.empty(1024)).download().reshape(1, -1)
    mine1 = np.array_split(np.arange(1), comm.size)[comm.rank]
    for mode in ("pipelined", "plain"):
        comm.force_pipeline = mode == "pipelined"
        s1 = sirt_mpi.SIRT(comm, geo1, p1.copy(), ang1, np.zeros((1, 3)), options={"_backend": HipBackend(sirt_mpi.SIRT._shard_geometry(geo1, mine1), ctx=ctx)})
        r1, e1 = s1.run_main_iteration(niter=3, positivity=True)
        out["one_angle_%s_rec" % mode], out["one_angle_%s_err" % mode] = r1, e1
    comm.force_pipeline = False
    out["one_angle_empty_ranks"] = np.array(int(round(comm.allreduce_scalar(1.0 if mine1.size == 0 else 0.0))))
    if comm.rank == 0:
        np.savez(out_path, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
