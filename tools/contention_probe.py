#!/usr/bin/env python3
"""One rank's share of the 8-GPU SIRT step (1024^3 volume, 128 of the 1024 angles, slab-pipelined reduce-scatter / all-gather on a ONE-rank RCCL
communicator = `bench.py --force-sharded --angles 128`) with and without the memory traffic the collectives of a real 8-rank run would add
beside the kernels.  On one rank RCCL moves nothing; the option comm_test_copy_eighths = 7 makes every asynchronous collective
copy 7/8 of the slab it touches on the communication stream -- per step 2 x 7/8 x 4.3 GB, the bytes a ring reduce-scatter + all-gather read
and write in one GPU's HBM at P = 8 -- either as a hipMemcpyAsync burst or through a copy kernel of w work-groups (RCCL-like: few work-groups
that hold CUs and stream at a limited rate; w sets the rate).
    python tools/contention_probe.py [N] [angles] [steps] [slabs] [quick]    -> a markdown table on stdout (profiles/round6_contention_probe.md)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
from tomography_alignment_amd import _lib  # noqa: E402
from tomography_alignment_amd.backend import HipBackend  # noqa: E402
from tomography_alignment_amd.comm import RcclComm  # noqa: E402
from tomography_alignment_amd.recon import sirt_mpi  # noqa: E402
from tomography_alignment_amd.utilities.geometry import Geometry  # noqa: E402
from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_proj = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
slabs = int(sys.argv[4]) if len(sys.argv) > 4 else None          # x slabs of the pipelined iteration (default: the solver's own, 8)
quick = len(sys.argv) > 5 and sys.argv[5] == "quick"             # only: none / 32 / 64 work-groups
NAMES = ("k_fwd_tile_flat", "k_adj_gather_flat", "k_fwd_live", "k_sino_zflags", "k_update", "k_residual_scale", "reduce_scatter_f32", "allgather_f32", "comm_join_wait")

ctx = _lib.Context(0)
comm = RcclComm(ctx, 0, 1, RcclComm.unique_id(ctx.lib))
comm.force_pipeline = True
geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
# the angles one rank of 8 owns: a contiguous eighth of linspace(0, pi, 8 * n_proj) (rank 3: generic, non axis-aligned directions)
phi = np.linspace(0.0, np.pi, 8 * n_proj)[3 * n_proj:4 * n_proj]
be = HipBackend(geo, ctx=ctx)
d_true = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
d_b = be.forward(_lib.poses_array(phi, 0 * phi, 0 * phi, np.zeros((n_proj, 3)), np.zeros(3)), d_true, be.empty(n_proj * N * N))
if slabs is not None:
    sirt_mpi.SIRT.n_pipeline_slabs = slabs
solver = sirt_mpi.SIRT(comm, geo, d_b, np.array([phi, 0 * phi, 0 * phi]).T, np.zeros((n_proj, 3)), {"_backend": be})
solver.iterate_device(niter=2)
ctx.sync()
slab_bytes = 4.0 * N ** 3
rows = []
for label, k8, wgs in (("no synthetic traffic (one-rank collectives move nothing)", 0, 0), ("hipMemcpyAsync burst", 7, 0), ("copy kernel, 8 work-groups", 7, 8),
                       ("copy kernel, 16 work-groups", 7, 16), ("copy kernel, 32 work-groups", 7, 32), ("copy kernel, 64 work-groups", 7, 64),
                       ("copy kernel, 256 work-groups", 7, 256), ("no synthetic traffic, again", 0, 0)):
    if quick and wgs not in (0, 32, 64) or (quick and label.startswith(("hipMemcpy", "no synthetic traffic, again"))):
        continue
    ctx.set_option("comm_test_copy_eighths", k8)
    ctx.set_option("comm_test_copy_wgs", wgs)
    solver.iterate_device(niter=1)
    ctx.sync()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    k_done, _ = solver.iterate_device(niter=steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    assert k_done == steps and solver._iter_pipelined
    ms = {nm: ctx.profile_get(nm)[1] / steps for nm in NAMES}
    coll = ms["reduce_scatter_f32"] + ms["allgather_f32"]
    copied = 2.0 * k8 / 8.0 * slab_bytes
    rows.append((label, 1e3 * dt / steps, ms, copied / 1e9, (2.0 * copied / (coll * 1e-3) / 1e9) if (k8 and coll > 0) else 0.0))
ctx.set_option("comm_test_copy_eighths", 0)
base = rows[0][1]
print("| synthetic traffic on the communication stream | step ms | vs none | k_fwd_tile_flat | k_adj_gather_flat | update + residual | reduce-scatter + all-gather (stream time) "
      "| comm_join_wait (exposed) | bytes copied per step | copy rate, read + write |")
print("|---|---|---|---|---|---|---|---|---|---|")
for label, step, ms, gb, rate in rows:
    print("| %s | %.1f | %+.1f %% | %.1f | %.1f | %.1f | %.1f | %.2f | %.1f GB | %s |"
          % (label, step, 100.0 * (step / base - 1.0), ms["k_fwd_tile_flat"], ms["k_adj_gather_flat"], ms["k_update"] + ms["k_residual_scale"],
             ms["reduce_scatter_f32"] + ms["allgather_f32"], ms["comm_join_wait"], gb, ("%.0f GB/s" % rate) if rate else "-"))
print()
print("%d^3 volume, %d angles (one rank's share of 8 x %d), %d timed steps per row, slab pipeline with %d slabs on a 1-rank RCCL communicator; kernel sources %s"
      % (N, n_proj, n_proj, steps, solver.n_pipeline_slabs, _lib.kernel_source_hash()))
comm.close()
