#!/usr/bin/env python3
"""
bench.py -- headline benchmark of BASELINE.json: SIRT iterations/s on a 1024^3 volume x 1024 angles
(parallel beam, Shepp-Logan, phi = linspace(0, pi)), with the dominant projector kernel priced against the
resource that bounds it.

    python bench.py --gpus N --steps K --warmup W          (N > 1: this script spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same result)

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment; torch is not imported).
A "step" is one SIRT iteration: A.rec, residual, A^T(W*res), all-reduce of the voxel update over the
angle shards (RCCL over xGMI; strong scaling: the 1024 angles are split across the N GPUs), update.
Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     the projector kernel that takes most of a step: work counted per launch by unit (VALU execution cycles, LDS-array
               cycles, HBM bytes -- rocprofv3 PMC passes of THIS command, committed under profiles/) / mean launch time
               measured live (HIP events on the kernel's own stream inside the timed region) against what the chip offers
               (MI355X_MICROARCH.md); `bound` is the unit with the highest utilisation.  The algorithmic-HBM figure of
               SURVEY 8d (volume re-read per angle) is kept as `hbm_algorithmic` -- it exceeds the HBM peak for the LDS-tile
               kernels, which read the volume from HBM once per CALL, and is therefore not the roofline.
  dense_volume the same SIRT step on a volume without zero regions (the tile kernels skip all-zero tiles; Shepp-Logan has many)
  cpu_baseline the CPU oracle (oracle/: plain-C port of the reference algorithm) timed on this box's host cores on a bounded
               sample of the same workload, extrapolated linearly in the number of rays / angles
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks (MI355X_MICROARCH.md: chip-level parameters; LDS table; wave scheduling)
HBM_PEAK_GBS = 8000.0         # HBM3E spec
N_CU = 256
CLK_GHZ = 2.4                 # max shader clock; the chip holds less under load, so fractions against it are conservative
VALU_PEAK_GINSTR = N_CU * 4 * CLK_GHZ / 2.0      # one wave64 VALU instruction per 2 cycles per SIMD-32, 4 SIMDs per CU = 1228.8 G/s
LDS_PEAK_GBS = N_CU * 128 * CLK_GHZ              # 128 B/clk/CU for ds_read_b32 / ds_read2_b32 / ds_read2st64_b32 = 78.6 TB/s ("~75 TB/s")
# ... and what tools/issue_bench.hip measured on this chip with every CU issuing (clock as held under that load): the
# practical ceilings, quoted beside the spec-derived ones
VALU_MEASURED_GINSTR = 570.0  # v_fma_f32 573, v_readlane 519, v_pk_fma_f32 475, v_add_u32 773 G wave-instr/s (profiles/round2_issue_bench.log)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md chip-level parameters
ATOMIC_PEAK_GBS = 1300.0          # global float atomics execute at the memory side at ~1.3 TB/s of added bytes chip-wide (MI355X_MICROARCH.md)
LDS_MEASURED_GINSTR2 = 142.0  # ds_read2_b32 / ds_read2st64_b32 / ds_add_u32 / ds_write_b32: 142 G wave-instr/s = 72.7 TB/s at 512 B each


def _count_gpus():
    """Number of visible GPUs, found in a CHILD process: the launcher itself must never initialise HIP (it starts other
    programs afterwards)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import ctypes\n"
            "from tomography_alignment_amd import _lib\n"
            "n = ctypes.c_int(0)\n"
            "rc = _lib.load().tomo_device_count(ctypes.byref(n))\n"
            "print(n.value if rc == 0 else 0)\n") % ROOT
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else 0
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return 0


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: spawn N fresh rank processes (one per GPU) BEFORE this process makes any
    HIP call, hand rank 0's JSON line through, exit non-zero if any rank does.  Replaces `mpirun -n N python ...` of the
    reference's recon/sirt_mpi.py:36-72 workflow."""
    import socket
    visible = _count_gpus()
    if visible < n:
        sys.stderr.write("bench.py: %d GPUs requested, %d visible\n" % (n, visible))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                TOMO_RDV_KEY="bench_%d_%d" % (os.getpid(), int(time.time() * 1e3) & 0xffffff), NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"))
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None))
    rc = 0
    deadline = time.time() + float(os.environ.get("TOMO_BENCH_LAUNCH_TIMEOUT", "3000"))
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                sys.stderr.write("bench.py: rank %d exited with code %s\n" % bad[0])
                rc = 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                sys.stderr.write("bench.py: ranks still running at the launcher's time limit\n")
                rc = 3
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()          # exactly the processes started here
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
    out0.seek(0)
    lines = [ln for ln in out0.read().decode(errors="replace").splitlines() if ln.strip()]
    if rc == 0 and lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
        return 0
    return rc or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="volume edge N (N^3 voxels, N x N detector)")
    ap.add_argument("--angles", type=int, default=1024)
    ap.add_argument("--perturbed", action="store_true", help="alpha,beta ~ U(+-1 deg), tx,tz ~ U(+-2 px) (default_rng(0))")
    ap.add_argument("--dense", action="store_true", help="headline on a volume without zero regions (no all-zero tile exits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-align", action="store_true", help="skip the alignment-gradient evals/s side measurement (config 5)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the align_rigid end-to-end side measurement (config 5: SIRT + one alignment pass)")
    ap.add_argument("--no-tilted", action="store_true", help="skip the tilted-pose SIRT side measurement")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-volume SIRT side measurement")
    ap.add_argument("--force-sharded", action="store_true",
                    help="N=1 only: run the multi-GPU code path (sharded solver, x-slab pipelined all-reduce) on a 1-rank RCCL communicator")
    ap.add_argument("--slabs", type=int, default=None, help="x slabs of the sharded solver's pipelined iteration (default: the solver's own)")
    ap.add_argument("--no-cgls", action="store_true", help="skip the CGLS side measurement")
    ap.add_argument("--no-shard-update", action="store_true",
                    help="sharded solver: all-reduce every slab and update the whole replica on every rank (round 3) instead of reduce-scatter -> "
                         "update of the rank's own 1/P -> all-gather")
    ap.add_argument("--only-align", choices=["near", "start", "dense"], default=None,
                    help="run ONLY that population of the alignment-gradient side measurement (config 5) and print its block: what the PMC passes of "
                         "tools/profile_round.sh profile (the last dispatch of each gradient kernel is then the timed one)")
    ap.add_argument("--detail", default=None, help="where the full measurement record goes (default: gpurun_out/bench_detail.json if that directory exists, "
                                                   "else bench_detail.json next to this file); stdout carries the contract line only")
    ap.add_argument("--fwd-variant", type=int, default=None)
    ap.add_argument("--adj-variant", type=int, default=None)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # nothing below has run: this process never touches the GPU

    # Libraries (RCCL prints a version banner at communicator init) must not pollute the ONE JSON line on stdout:
    # send everything written to fd 1 during the run to stderr and keep the real stdout for the final line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.comm import RcclComm
    from tomography_alignment_amd.recon import sirt as sirt_mod, sirt_mpi, cgls as cgls_mod, cgls_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry
    from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    N, n_proj = args.size, args.angles
    comm = RcclComm.from_env()
    ctx = comm.ctx
    if args.force_sharded and world == 1:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        comm = RcclComm(ctx, 0, 1, RcclComm.unique_id(ctx.lib))
        comm.force_pipeline = True
    if args.fwd_variant is not None:
        ctx.set_option("fwd_variant", args.fwd_variant)
    if args.adj_variant is not None:
        ctx.set_option("adj_variant", args.adj_variant)

    if args.only_align:
        out = {"alignment_gradient": align_rate(comm, ctx, rank, world, N=min(512, max(32, N // 2)), n_proj=720 if N >= 1024 else max(8, n_proj // 2),
                                                legs=(args.only_align,))}
        if rank == 0:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if world > 1 or args.force_sharded:
            comm.close()
        return

    geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    phi = np.linspace(0., np.pi, n_proj)

    def poses_for(tilted):
        alpha, beta, xyz = np.zeros(n_proj), np.zeros(n_proj), np.zeros((n_proj, 3))
        if tilted:
            rng = np.random.default_rng(0)
            alpha = np.deg2rad(rng.uniform(-1, 1, n_proj))
            beta = np.deg2rad(rng.uniform(-1, 1, n_proj))
            xyz[:, 0] = rng.uniform(-2, 2, n_proj)
            xyz[:, 2] = rng.uniform(-2, 2, n_proj)
        return alpha, beta, xyz

    # ---- synthetic data, generated and kept on the device: phantom -> this rank's sinogram rows
    my_rows = np.array_split(np.arange(n_proj), world)[rank]
    shard_geo = sirt_mpi.SIRT._shard_geometry(geo, my_rows)
    be = HipBackend(shard_geo, ctx=ctx)
    d_true = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
    d_b = be.empty(my_rows.size * N * N)
    opts = {"_backend": be}

    def make_dense(vol):
        """Shepp-Logan + 0.05 everywhere: same object, no voxel exactly zero (the zero-tile exits never fire)."""
        one = be.empty(vol.size)
        be.fill(one, 0.05)
        be.axpy(vol, one, 1.0)
        del one

    if args.slabs is not None:
        sirt_mpi.SIRT.n_pipeline_slabs = int(args.slabs)
    if args.no_shard_update:
        sirt_mpi.SIRT.shard_update = False

    def make_solver(tilted):
        alpha, beta, xyz = poses_for(tilted)
        poses = _lib.poses_array(phi[my_rows], alpha[my_rows], beta[my_rows], xyz[my_rows], np.zeros(3))
        be.forward(poses, d_true, d_b)
        angles = np.array([phi, alpha, beta]).T
        if world > 1 or args.force_sharded:
            return sirt_mpi.SIRT(comm, geo, d_b, angles, xyz, opts)
        return sirt_mod.SIRT(geo, d_b, angles, xyz, opts)

    def barrier():
        ctx.sync()
        comm.barrier()
        ctx.sync()

    if args.dense:
        make_dense(d_true)
    solver = make_solver(args.perturbed)
    if args.warmup > 0:
        solver.iterate_device(niter=args.warmup)
    barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    k_done, rms = solver.iterate_device(niter=args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    elapsed = comm.allreduce_max(elapsed)
    if k_done != args.steps:
        raise SystemExit("bench.py: solver stopped after %d of %d steps (semi-convergence rule fired)" % (k_done, args.steps))

    # ---- per-kernel timing of the timed region (HIP events on the stream each kernel / collective runs on)
    kern = {}
    for name in ("k_fwd_v1", "k_fwd_v2", "k_fwd_tile", "k_fwd_tile_flat", "k_fwd_live", "k_adj_v1", "k_adj_tile", "k_adj_tile_flat", "k_adj_gather_flat", "k_pad", "k_unpad",
                 "k_absmax", "k_sino_zflags", "k_residual_scale", "k_update", "k_vec", "allreduce_f32", "reduce_scatter_f32", "allgather_f32", "comm_join_wait"):
        n, ms = ctx.profile_get(name)
        if n:
            # a pass over all angles may be issued as several launches (x slabs of the pipelined back-projection):
            # ms_per_step sums them, so work-per-pass / ms_per_step == work-per-launch / avg launch time
            kern[name] = {"launches": n, "avg_ms": ms / n, "launches_per_step": n / float(args.steps), "ms_per_step": ms / float(args.steps)}
    # bytes each rank hands to the collectives per step (the voxel update, float32) and what a ring moves per rank for them: an
    # all-reduce of the volume, or (round 4: the update sharded over the ranks) its two halves -- reduce-scatter, then all-gather
    for cname, ring in (("allreduce_f32", 2.0), ("reduce_scatter_f32", 1.0), ("allgather_f32", 1.0)):
        if cname in kern:
            kern[cname]["bytes_per_step"] = 4.0 * N ** 3
            kern[cname]["ring_bytes_per_rank_per_step"] = ring * (world - 1) / world * 4.0 * N ** 3
    comm_names = [c for c in ("allreduce_f32", "reduce_scatter_f32", "allgather_f32") if c in kern]
    if comm_names:
        total = sum(kern[c]["ms_per_step"] for c in comm_names)
        kern[comm_names[0]]["exposed_ms_per_step"] = kern.get("comm_join_wait", {}).get("ms_per_step", total)      # of all collectives of the step together
    n_loc = my_rows.size
    n_det = N * N
    alg_fwd = n_loc * (4.0 * N ** 3 + 4.0 * n_det)               # bytes per forward launch   
    alg_adj = n_loc * (8.0 * N ** 3 + 4.0 * n_det)               # bytes per back-projection launch
    def roofline_pair(kk, key):
        """(roofline of the projector kernel that takes most of a step, the other one's, extras) from a {kernel: {ms_per_step, launches_per_step}} table."""
        fwd = next((k for k in ("k_fwd_tile_flat", "k_fwd_tile", "k_fwd_v2", "k_fwd_v1") if k in kk), None)
        adj = next((k for k in ("k_adj_gather_flat", "k_adj_tile_flat", "k_adj_tile", "k_adj_v1") if k in kk), None)
        cands = []
        if fwd:
            cands.append((kk[fwd]["ms_per_step"], fwd, alg_fwd))
        if adj:
            cands.append((kk[adj]["ms_per_step"], adj, alg_adj))
        main_r, other_r, ex = None, None, {}
        if cands:
            step_ms, name, alg = max(cands)
            main_r = make_roofline(name, step_ms, kk[name]["launches_per_step"], alg, key, n_loc * float(N) ** 3, 4.0 * n_loc * n_det)
        if fwd:
            ex["forward_alg_GBps"] = round(alg_fwd / (kk[fwd]["ms_per_step"] * 1e-3) / 1e9, 1)
        if adj:
            ex["backproj_alg_GBps"] = round(alg_adj / (kk[adj]["ms_per_step"] * 1e-3) / 1e9, 1)
        if main_r is not None and len(cands) == 2:      # the other projector kernel, priced the same way (secondary)
            o_ms, o_name, o_alg = min(cands)
            other_r = make_roofline(o_name, o_ms, kk[o_name]["launches_per_step"], o_alg, key, n_loc * float(N) ** 3, 4.0 * n_loc * n_det)
        return main_r, other_r, ex, fwd, adj

    base_key = "N%d_A%d_G%d" % (N, n_proj, world)
    key = base_key + ("_P" if args.perturbed else "") + ("_D" if args.dense else "")
    roofline, other, extra, fwd_name, adj_name = roofline_pair(kern, key)
    if other is not None:
        extra["roofline_other_kernel"] = other

    its = args.steps / elapsed
    out = {
        "metric": "sirt_iterations_per_sec",
        "value": round(its, 6),
        "unit": "it/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (3-D Shepp-Logan generated on the device%s; sinogram = its forward projection)" % (" + 0.05 everywhere" if args.dense else ""),
        "config": {"workload": "SIRT %d^3 volume x %d angles, parallel beam, step 1.0, detector %dx%d%s%s"
                               % (N, n_proj, N, N, ", perturbed poses" if args.perturbed else "", ", dense volume" if args.dense else ""),
                   "sharding": "angles split over %d GPU(s), RCCL all-reduce of the voxel update" % world,
                   "rms_error_last": float(rms[-1])},
        "roofline": roofline,
        "kernels": kern,
    }
    out.update(extra)
    if roofline is not None and fwd_name and adj_name:
        step_alg = n_proj * (12.0 * N ** 3 + 8.0 * n_det) + 16.0 * n_proj * n_det + 16.0 * N ** 3
        out["sirt_step_alg_GBps"] = round(step_alg / (elapsed / args.steps) / 1e9, 1)

    def side_run(tilted, dense, label):
        """2 timed iterations (1 warm-up) of the same workload with other poses / another volume: {value, unit, ...}."""
        nonlocal solver
        del solver
        if dense:
            make_dense(d_true)
        solver = make_solver(tilted)
        solver.iterate_device(niter=1)
        barrier()
        ctx.profile_reset()
        ctx.profile_enable(True)
        t1 = time.perf_counter()
        solver.iterate_device(niter=2)
        barrier()
        dt = comm.allreduce_max(time.perf_counter() - t1)
        ctx.profile_enable(False)
        kk, table = {}, {}
        for nm in ("k_fwd_tile", "k_fwd_tile_flat", "k_adj_tile", "k_adj_tile_flat", "k_adj_gather_flat"):
            n, ms = ctx.profile_get(nm)
            if n:
                kk[nm + "_ms_per_step"] = round(ms / 2.0, 2)
                table[nm] = {"launches": n, "avg_ms": ms / n, "launches_per_step": n / 2.0, "ms_per_step": ms / 2.0}
        # everything else a step launches -- the work lists rebuilt by every call (k_fwd_live: block / tile classification + compaction;
        # k_sino_zflags: non-empty sinogram planes), the fixed-point scale (k_absmax), the vector lines -- and what the wall clock holds beyond
        # the kernels (host-side staging of the pose constants, launch gaps): VERDICT r5 next 7 asked what caching the lists could gain
        for nm in ("k_fwd_live", "k_sino_zflags", "k_absmax", "k_pad", "k_unpad", "k_residual_scale", "k_update", "k_vec"):
            n, ms = ctx.profile_get(nm)
            if n:
                kk[nm + "_ms_per_step"] = round(ms / 2.0, 2)
        kk["wall_minus_kernels_ms_per_step"] = round(1e3 * dt / 2.0 - sum(v for k, v in kk.items() if k.endswith("_ms_per_step")), 2)
        # the leg's own roofline blocks (VERDICT r3 #2b), from counters committed for ITS workload key (..._P tilted poses, ..._D dense volume;
        # tools/profile_round.sh takes the PMC passes of `bench.py --perturbed` / `--dense`, the same solver on the same data)
        leg_key = base_key + ("_P" if tilted else "") + ("_D" if dense else "")
        main_r, other_r, _, _, _ = roofline_pair(table, leg_key)
        res = dict({"value": round(2.0 / dt, 6), "unit": "it/s", "steps": 2, "warmup": 1, "config": label}, **kk)
        if main_r is not None:
            res["roofline"] = main_r
        if other_r is not None:
            res["roofline_other_kernel"] = other_r
        return res

    if not args.no_tilted and not args.perturbed and not args.dense:
        # side measurement (not `value`): alpha, beta ~ U(+-1 deg), tx, tz ~ U(+-2 px), default_rng(0) -- SURVEY 8d's perturbed run:
        # these take the general tile kernels, which is what SIRT runs on once an alignment pass has moved the poses
        out["tilted_poses"] = side_run(True, False, "same workload, alpha, beta ~ U(+-1 deg), tx, tz ~ U(+-2 px): general tile kernels")
    if not args.no_dense and not args.dense:
        # side measurement: the same object + 0.05 everywhere, so that no tile is all zero (the headline leans on
        # the zero-tile exits of the tile kernels; Shepp-Logan is exactly zero outside its ellipsoid)
        out["dense_volume"] = side_run(args.perturbed, True, "same workload on a volume with no zero voxel (Shepp-Logan + 0.05): no all-zero tile exits")
    if not args.no_cgls and not args.perturbed and not args.dense:
        # side measurement: CGLS iterations per second on the same workload, on every world size.  One GPU, plain:
        # recon/cgls.py:54-82 -- two forward projections (A p, and A rec for its ||b - A rec|| restart test) and one back-projection per
        # iteration.  Sharded (world > 1 or --force-sharded): recon/cgls_mpi.py:70-107 -- its monitor is ||b - A p||, so ONE forward and one
        # back-projection per iteration; slab pipeline as the sharded SIRT's (reduce-scatter per slab, gamma accumulated on the device, p
        # updated piecewise and all-gathered with the next A p behind the all-gathers), two small device-side scalar all-reduces
        del solver
        solver = None
        sharded = world > 1 or args.force_sharded
        be.phantom(d_true, (N, N, N), SHEPP_LOGAN)          # the dense leg above added 0.05 in place
        be.forward(_lib.poses_array(phi[my_rows], 0 * phi[my_rows], 0 * phi[my_rows], np.zeros((my_rows.size, 3)), np.zeros(3)), d_true, d_b)
        c_args = (geo, d_b, np.array([phi, 0 * phi, 0 * phi]).T, np.zeros((n_proj, 3)), {"_backend": be})
        c = cgls_mpi.CGLS(comm, *c_args) if sharded else cgls_mod.CGLS(*c_args)
        c.iterate_device(niter=1)
        barrier()
        ctx.profile_reset()
        ctx.profile_enable(True)
        t1 = time.perf_counter()
        k_c, rms_c = c.iterate_device(niter=3)
        barrier()
        dt = comm.allreduce_max(time.perf_counter() - t1)
        ctx.profile_enable(False)
        kk = {}
        for nm in ("k_fwd_tile_flat", "k_adj_gather_flat", "k_fwd_live", "k_sino_zflags", "k_vec", "k_dot", "reduce_scatter_f32", "allgather_f32", "allreduce_f32",
                   "allreduce_scalars", "comm_join_wait"):
            n, ms = ctx.profile_get(nm)
            if n:
                kk[nm + "_ms_per_step"] = round(ms / float(k_c), 2)
        out["cgls"] = dict({"value": round(k_c / dt, 6), "unit": "it/s", "steps": int(k_c), "warmup": 1, "rms_error_last": float(rms_c[-1]),
                            "pipelined": bool(getattr(c, "_pipelined", False)),
                            "config": ("CGLS, angle-sharded over %d rank(s) (recon/cgls_mpi.py:70-107): 1 forward + 1 back-projection per iteration, slab pipeline" % world) if sharded
                            else "CGLS (recon/cgls.py:54-82) on the same workload: 2 forward projections + 1 back-projection per iteration"}, **kk)
        out["cgls_it_per_s"] = out["cgls"]["value"]
        del c
    # the rates of the side legs beside `value` (VERDICT r3 #2c): `value` is measured on the Shepp-Logan phantom, a quarter of whose
    # blocks are all zero and never launched; `value_dense_volume` does all the work; `value_tilted_poses` is what SIRT runs at once an
    # alignment pass has moved the poses
    if "dense_volume" in out:
        out["value_dense_volume"] = out["dense_volume"]["value"]
    if "tilted_poses" in out:
        out["value_tilted_poses"] = out["tilted_poses"]["value"]
    if not args.no_align:
        del solver
        out["alignment_gradient"] = align_rate(comm, ctx, rank, world, N=min(512, max(32, N // 2)), n_proj=720 if N >= 1024 else max(8, n_proj // 2))
    if not args.no_align and not args.no_e2e:
        # config 5 end to end: two outer iterations of examples/align_rigid.py's loop -- on every world size (every rank
        # runs the same legs in the same order, rank 0 prints); world > 1 or --force-sharded take the sharded code path
        out["align_rigid_e2e"] = align_rigid_e2e(comm, ctx, rank, world, sharded=(world > 1 or args.force_sharded),
                                                 N=min(512, max(32, N // 2)), n_proj=720 if N >= 1024 else max(8, n_proj // 2))
    if roofline is not None:
        roofline["measured_d2d_copy_GBps"] = copy_probe(ctx, be)        # read + write of a 2 GiB hipMemcpy D2D, same run
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        d_ref = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
        out["cpu_baseline"] = cpu_baseline(be, d_ref, N, n_proj, phi)
    if rank == 0:
        emit(out, real_stdout, args.detail)
    if world > 1 or args.force_sharded:
        comm.close()


# ---- what goes where.  The driver parses rank 0's LAST stdout line out of a bounded tail of stdout: round 5's
# line had grown to 25 KB (whole PMC dictionaries, per-outer tables) and was cut in front, so nothing of it was parsed.  stdout carries
# the CONTRACT only (contract_line(), < LINE_LIMIT bytes, held there by tests/test_host_logic.py at the full-size counters);
# everything measured goes to bench_detail.json next to this file (gpurun_out/ on a GPU box, if that is writable) and to stderr.
LINE_LIMIT = 4096
_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_per_step", "traffic", "traffic_source",
              "hbm_algorithmic_frac", "hbm_counter_frac", "atomics_frac", "useful_flop_frac", "measured_d2d_copy_GBps")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _sig(v, digits=5):
    """floats at `digits` significant digits (the line is a summary; bench_detail.json has what was measured)."""
    if isinstance(v, float) and v == v and v not in (float("inf"), float("-inf")):
        return float("%.*g" % (digits, v))
    return v


def compact_roofline(r):
    if not isinstance(r, dict):
        return None
    c = {k: _sig(r[k]) for k in _ROOF_KEYS if k in r}
    if "unit" in c:
        c["unit"] = _short(c["unit"], 40)
    if "kernels" in r and "kernel" not in c:                     # the gradient legs price one pass of several launches
        c["kernel"] = "+".join(r["kernels"])
        c["avg_launch_ms"] = r.get("ms_per_pass")
    if c.get("traffic_source"):
        # "profiles/pmc_traffic.json (round5, workload N1024_A1024_G1, kernel sources 839ec0ddf518aba0)": keep file, round and source hash
        c["traffic_source"] = _short(c["traffic_source"], 110)
    if r.get("counters") is None and "bound" in c:
        c["counters"] = None                                     # says: algorithmic-HBM fallback, no committed PMC passes for this workload / these sources
    if r.get("stale_counters_refused"):
        c["stale_counters_refused"] = len(r["stale_counters_refused"])
    if c.get("hbm_algorithmic_frac") is not None and c["hbm_algorithmic_frac"] > 1.0 and c.get("bound") != "hbm":
        c["note"] = ("bound / frac = the busiest MEASURED unit (PMC counters / live time); hbm_algorithmic_frac = SURVEY 8d bytes (volume re-read per angle) "
                     "/ time / 8 TB/s, above 1 because a block is staged once for all angles; hbm_counter_frac = what the HBM carried")
    return c


def contract_line(out):
    """The ONE stdout line: the driver's contract keys, a compact `roofline` and `cpu_baseline`, the side legs as scalars."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype")}
    line["data"] = _short(out.get("data", "synthetic"), 120)
    cfg = out.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload", ""), 160), "sharding": _short(cfg.get("sharding", ""), 100)}
    line["roofline"] = compact_roofline(out.get("roofline"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": _short(cb.get("sample", ""), 230)}
        if isinstance(cb.get("all_cores"), dict):
            c["all_cores"] = {"value": cb["all_cores"].get("value"), "cores": cb["all_cores"].get("cores")}
        ref = cb.get("reference")
        if isinstance(ref, dict):
            c["reference"] = {"kind": "reference", "available": bool(ref.get("available")), "cores": ref.get("cores"),
                              "forward": {"s_per_angle": _sig(_get(ref, "forward", "s_per_angle"))},
                              "gradient": {"s_per_eval": _sig(_get(ref, "gradient", "s_per_eval"))}}
        g = cb.get("gradient")
        if isinstance(g, dict):
            c["gradient"] = {"evals_per_sec": g.get("evals_per_sec"), "cores": g.get("cores"), "all_cores_evals_per_sec": _get(g, "all_cores", "evals_per_sec")}
        line["cpu_baseline"] = c
    for k in ("value_dense_volume", "value_tilted_poses", "cgls_it_per_s", "forward_alg_GBps", "backproj_alg_GBps", "sirt_step_alg_GBps"):
        if out.get(k) is not None:
            line[k] = out[k]
    ro = out.get("roofline_other_kernel")
    if isinstance(ro, dict):
        line["roofline_other_kernel"] = {k: _sig(ro[k]) if k != "unit" else _short(ro[k], 40) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "hbm_algorithmic_frac", "hbm_counter_frac") if k in ro}
    ag = out.get("alignment_gradient")
    if isinstance(ag, dict):
        line["alignment_gradient"] = {"evals_per_sec": ag.get("evals_per_sec"), "unit": ag.get("unit"), "evals_per_sec_dense_volume": _get(ag, "dense_volume", "evals_per_sec"),
                                      "evals_per_sec_at_untilted_start": ag.get("evals_per_sec_at_untilted_start"),
                                      "roofline": {k: v for k, v in (compact_roofline(ag.get("roofline")) or {}).items() if k in ("kernel", "bound", "frac", "avg_launch_ms", "hbm_algorithmic_frac", "hbm_counter_frac", "counters")} or None}
    e = out.get("align_rigid_e2e")
    if isinstance(e, dict):
        line["align_rigid_e2e"] = {"wall_s": e.get("wall_s"), "outer_iterations": len(e.get("outer", [])), "sirt_wall_s_flat_poses": _sig(e.get("sirt_wall_s_flat_poses")),
                                   "sirt_wall_s_recovered_poses": _sig(e.get("sirt_wall_s_recovered_poses")), "align_wall_s": _sig(e.get("align_wall_s")),
                                   "evals_per_sec_end_to_end": e.get("evals_per_sec_end_to_end")}
    kern = out.get("kernels")
    if isinstance(kern, dict):
        # per-step milliseconds of the timed region by kernel / collective (HIP events): kernel time <= step time can be checked from the line
        line["kernel_ms_per_step"] = {k: round(v["ms_per_step"], 2) for k, v in kern.items() if v.get("ms_per_step", 0.0) >= 0.05}
    if out.get("detail_file"):
        line["detail"] = out["detail_file"]
    s = json.dumps(line, separators=(", ", ": "))
    if len(s) >= LINE_LIMIT:                                     # never let an added leg push the contract keys out of the driver's tail again
        for k in ("kernel_ms_per_step", "roofline_other_kernel", "align_rigid_e2e", "alignment_gradient"):
            line.pop(k, None)
            s = json.dumps(line, separators=(", ", ": "))
            if len(s) < LINE_LIMIT:
                break
    return s


def emit(out, stdout_fd, detail_path=None):
    """Detail to bench_detail.json (+ stderr), the contract line -- and only it -- to the real stdout."""
    detail = json.dumps(out, indent=1, sort_keys=False)
    for d in ([detail_path] if detail_path else []) + [os.path.join(ROOT, "gpurun_out"), ROOT, "/tmp"]:
        try:
            if d.endswith("gpurun_out") and not os.path.isdir(d):
                continue
            path = d if d == detail_path else os.path.join(d, "bench_detail.json")
            with open(path, "w") as f:
                f.write(detail + "\n")
            out["detail_file"] = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
            break
        except OSError:
            continue
    sys.stdout.flush()                                           # fd 1 is stderr here (main() re-pointed it)
    sys.stderr.write("bench.py detail (also in %s):\n%s\n" % (out.get("detail_file"), json.dumps(out)))
    sys.stderr.flush()
    os.write(stdout_fd, (contract_line(out) + "\n").encode())


def load_counters(fname, key, kernel, stale):
    """One kernel's entry of a committed counter file (profiles/sq_counters.json, profiles/pmc_traffic.json) for the workload `key`,
    or None: files hold {"source", "src_hash", "workloads": {key: {"workload", "kernels": {...}}}} (round 4; a round-3 file has one
    "key" / "kernels" pair at the top).  Counters taken on other kernel sources are refused and named in `stale`."""
    from tomography_alignment_amd import _lib
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", fname)))
        w = j["workloads"].get(key) if "workloads" in j else ({"kernels": j["kernels"]} if j.get("key") == key else None)
        if w is None or kernel not in w["kernels"]:
            return None, None
        live_hash = _lib.kernel_source_hash()
        if j.get("src_hash") != live_hash:
            stale.append("profiles/%s (%s) was taken on kernel sources %s, this run has %s" % (fname, j.get("source", ""), j.get("src_hash"), live_hash))
            return None, None
        return w["kernels"][kernel], "profiles/%s (%s, workload %s, kernel sources %s)" % (fname, j.get("source", ""), key, live_hash)
    except (OSError, ValueError, KeyError):
        return None, None


def make_roofline(name, ms_per_step, launches_per_step, alg_bytes_per_pass, key, useful_samples_per_pass=None, useful_sino_bytes=None):
    """Utilisation of every unit that could bound `name`, from counted work per launch (committed rocprofv3 PMC passes of this
    very command, profiles/sq_counters.json + profiles/pmc_traffic.json) and the live launch time; `bound` = the busiest unit.
    Without counters for this workload: the algorithmic-HBM figure only, flagged `counters: null`."""
    t = ms_per_step * 1e-3                                   # seconds per pass over all angles (sums the x-slab launches)
    alg_gbs = alg_bytes_per_pass / t / 1e9
    r = {"kernel": name, "avg_launch_ms": round(ms_per_step / launches_per_step, 3), "launches_per_step": launches_per_step,
         "hbm_algorithmic": {"bytes_per_launch": alg_bytes_per_pass / launches_per_step, "GBps": round(alg_gbs, 1), "frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
                             "note": "SURVEY 8d byte model (volume re-read per angle); the LDS-tile kernels read the volume once per call, so this is not a roofline for them"},
         "traffic": None, "counters": None}
    sq, pmc, write_bytes = None, None, None
    # Committed counters are used only for THIS workload (key) taken on THESE kernel sources (src_hash over csrc/*): counts of an
    # older kernel divided by the live time of a newer one would be a silent mix.
    stale = []
    ent, src = load_counters("sq_counters.json", key, name, stale)
    if ent is not None:
        sq = ent
        r["counters"] = {"source": src, "per_pass": sq}
    ent, src = load_counters("pmc_traffic.json", key, name, stale)
    if ent is not None:
        pmc = ent["hbm_bytes_per_launch"]
        write_bytes = ent.get("write_kb", 0.0) * 1024.0
        r["traffic"] = pmc / launches_per_step
        r["traffic_source"] = src
    if stale:
        r["stale_counters_refused"] = stale
    # useful arithmetic: every in-volume ray sample needs the four x,y-corner FMAs of its plane pair's bilinear weights = 8 flop in
    # the regrouped (flat) form -- n_samples = n_angles * N^3 per pass (a unit lattice has one sample per voxel volume) -- against
    # the fp32 vector peak.  Independent of how many instructions the kernel spends per sample: the busy fractions below say how
    # full the pipes are, this says how much of what they do is the interpolation itself.
    if useful_samples_per_pass:
        r["useful_flop_frac"] = round(8.0 * useful_samples_per_pass / t / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)
        r["useful_flop"] = {"samples_per_pass": useful_samples_per_pass, "flop_per_sample": 8, "TFLOPs": round(8.0 * useful_samples_per_pass / t / 1e12, 2),
                            "peak_TFLOPs": FP32_VECTOR_PEAK_TFLOPS}
    if write_bytes is not None and name.startswith("k_fwd"):
        # the forward kernels' stores are all float atomics into the sinogram (WRITE_SIZE is exact for them, MI355X_MICROARCH.md)
        r["atomics_frac"] = round(write_bytes / t / 1e9 / ATOMIC_PEAK_GBS, 4)
        r["atomics"] = {"bytes_per_pass": write_bytes, "GBps": round(write_bytes / t / 1e9, 1), "memory_side_ceiling_GBps": ATOMIC_PEAK_GBS,
                        "amplification_vs_sinogram": round(write_bytes / max(1.0, useful_sino_bytes), 1) if useful_sino_bytes else None}
    util = {}
    if pmc is not None:
        util["hbm"] = (pmc / t / 1e9, HBM_PEAK_GBS, "GB/s")
    if write_bytes is not None and name.startswith("k_fwd"):
        # a unit of its own: no-return float atomics execute at the memory side at ~1.3 TB/s of added bytes whatever their locality or
        # scope (MI355X_MICROARCH.md; tools/gatomic_scope_bench.hip) -- the forward kernels cannot finish before their atomics have
        util["atomics"] = (write_bytes / t / 1e9, ATOMIC_PEAK_GBS, "GB/s of float atomics")
    if sq is not None:
        # unit-busy cycles counted by the SQ for this launch (MI355X_MICROARCH: SQ_LDS_IDX_ACTIVE = all LDS-array cycles;
        # SQ_ACTIVE_INST_VALU counts quad-cycles of VALU execution per SIMD) against the cycles the chip has in the live time
        if sq.get("SQ_ACTIVE_INST_VALU"):
            util["valu"] = (4.0 * sq["SQ_ACTIVE_INST_VALU"] / t / 1e9, N_CU * 4 * CLK_GHZ, "G SIMD-cycles/s")
        if sq.get("SQ_LDS_IDX_ACTIVE"):
            util["lds"] = (sq["SQ_LDS_IDX_ACTIVE"] / t / 1e9, N_CU * CLK_GHZ, "G LDS-cycles/s")
    # The two HBM fractions side by side, at the top of the block (VERDICT r3 #2a): the SURVEY 8(d) figure -- algorithmic bytes (the
    # volume re-read for every angle) / live time / 8 TB/s, which EXCEEDS 1 for a kernel that stages a tile once for all angles -- and
    # what the HBM actually carried according to the counters (null without counters for this workload and these sources).
    r["hbm_algorithmic_frac"] = round(alg_gbs / HBM_PEAK_GBS, 4)
    r["hbm_counter_frac"] = round(pmc / t / 1e9 / HBM_PEAK_GBS, 4) if pmc is not None else None
    if not util:
        # no counters for this exact workload: fall back to the algorithmic-HBM figure, capped reading left to the consumer
        r.update({"bound": "hbm", "achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / HBM_PEAK_GBS, 4),
                  "note": "no committed PMC counters for workload key %s: algorithmic-HBM figure (may exceed 1 for the LDS-tile kernels)" % key})
        return r
    # `valu` is AMD's VALUBusy: a wave64 VALU instruction is counted as one quad-cycle of its SIMD although the SIMD retires the
    # simple integer ones faster (tools/issue_bench.hip: v_add_u32 one per 2 clk, v_fma_f32 one per 2.65), so a kernel made of those
    # can read slightly above 1 (the tilted adjoint reads 1.04); `lds`, `hbm` and `atomics` are capped by 1.
    bound = max(util, key=lambda k: util[k][0] / util[k][1])
    ach, peak, unit = util[bound]
    r.update({"bound": bound, "achieved": round(ach, 1), "peak": round(peak, 1), "unit": unit, "frac": round(ach / peak, 4),
              "utilisation": {k: {"achieved": round(v[0], 1), "peak": round(v[1], 1), "unit": v[2], "frac": round(v[0] / v[1], 4)} for k, v in util.items()}})
    if sq is not None:
        # the same launch in instruction counts (secondary): wave-instructions per second by unit, and the LDS bytes they move
        # (SQ_INSTS_LDS x the bytes of this kernel's LDS instruction, tools/summarise_sq.py) against the spec / measured ceilings
        info = {}
        if sq.get("SQ_INSTS_VALU"):
            info["valu_Ginstr_per_s"] = round(sq["SQ_INSTS_VALU"] / t / 1e9, 1)
            info["valu_frac_of_simd32_issue_peak"] = round(sq["SQ_INSTS_VALU"] / t / 1e9 / VALU_PEAK_GINSTR, 4)
            info["valu_frac_of_measured_fma_ceiling"] = round(sq["SQ_INSTS_VALU"] / t / 1e9 / VALU_MEASURED_GINSTR, 4)
        if sq.get("lds_bytes"):
            info["lds_GBps"] = round(sq["lds_bytes"] / t / 1e9, 1)
            info["lds_frac_of_peak_bandwidth"] = round(sq["lds_bytes"] / t / 1e9 / LDS_PEAK_GBS, 4)
        if sq.get("SQ_INSTS_SALU"):
            info["salu_Ginstr_per_s"] = round(sq["SQ_INSTS_SALU"] / t / 1e9, 1)
            # one scalar unit per CU: the gather back-projection was limited by it at 0.6 of one instruction per clock (HISTORY.md section 4)
            info["salu_frac_of_one_per_clk_per_cu"] = round(sq["SQ_INSTS_SALU"] / t / 1e9 / (N_CU * CLK_GHZ), 4)
        r["instruction_rates"] = info
    return r


def align_rate(comm, ctx, rank, world, N=512, n_proj=720, passes=3, legs=("near", "start", "dense")):
    """Side measurement (not `value`): alignment cost+gradient evaluations per second on BASELINE config 5 -- 512^3
    volume, 720 projections with +-2 deg / +-5 px perturbations (default_rng(5)); projections are sharded over the
    ranks with a replicated volume and no collective; one fused launch evaluates a rank's whole shard.
    Every population carries a `roofline` block: the gradient kernels' live launch times (HIP events) against the work the committed
    PMC passes of the same command counted (profiles/sq_counters.json / pmc_traffic.json, workload keys C5_N<N>_P<n>_G<w>_<leg>)."""
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.utilities.geometry import Geometry
    from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN
    rng = np.random.default_rng(5)
    phi = np.linspace(0., np.pi, n_proj)
    alpha, beta = np.deg2rad(rng.uniform(-2, 2, n_proj)), np.deg2rad(rng.uniform(-2, 2, n_proj))
    xyz = np.zeros((n_proj, 3))
    xyz[:, 0], xyz[:, 2] = rng.uniform(-5, 5, n_proj), rng.uniform(-5, 5, n_proj)
    mine = np.array_split(np.arange(n_proj), world)[rank]
    geo = Geometry(mine.size, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    be = HipBackend(geo, ctx=ctx)
    vol = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
    truth = _lib.poses_array(phi[mine], alpha[mine], beta[mine], xyz[mine], np.zeros(3))
    b = be.forward(truth, vol, be.empty(mine.size * N * N))
    # An optimiser starts at the nominal (untilted) poses and ends near the true (tilted) ones; the kernels are slower on
    # tilted poses, so the headline rate is taken at poses half a degree / one pixel away from the truth and the rate at
    # the untilted start is reported beside it.
    near = _lib.poses_array(phi[mine], alpha[mine] + np.deg2rad(0.5), beta[mine] - np.deg2rad(0.5), xyz[mine] + np.array([1.0, 0.0, -1.0]), np.zeros(3))
    start = _lib.poses_array(phi[mine], 0 * alpha[mine], 0 * beta[mine], 0 * xyz[mine], np.zeros(3))
    alg = 4.0 * N ** 3 + 4.0 * N * N + 28.0                      # fused form, BASELINE.md section 3

    def rate(poses, volume, leg):
        be.cost_grad(poses, volume, b)
        ctx.sync()
        comm.barrier()
        ctx.profile_reset()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(passes):
            cost, g6 = be.cost_grad(poses, volume, b)
        ctx.sync()
        comm.barrier()
        dt = comm.allreduce_max(time.perf_counter() - t0)
        ctx.profile_enable(False)
        # the launches of one pass (one per kernel variant in use: poses are grouped by tilt), priced together
        per = {}
        for v in ("k_cost_grad(v1)", "k_cost_grad(v2)", "k_cost_grad(v3)"):
            n, ms = ctx.profile_get(v)
            if n:
                per[v] = {"launches_per_pass": n / float(passes), "ms_per_pass": round(ms / passes, 3)}
        roof = grad_roofline(per, mine.size * alg, "C5_N%d_P%d_G%d_%s" % (N, n_proj, world, leg), mine.size * float(N) ** 3)
        return passes * n_proj / dt, float(cost[0]), per, roof

    out = {"unit": "evals/s", "projections_per_launch": int(mine.size)}
    if "near" in legs:
        r_near, c_near, per, roof = rate(near, vol, "near")
        out.update({"evals_per_sec": round(r_near, 1), "config": "%d^3 volume, %d projections simulated with +-2 deg / +-5 px pose "
                    "errors, fused cost+6-DoF gradient evaluated 0.5 deg / 1 px away from the true poses" % (N, n_proj),
                    "alg_GBps": round(r_near * alg / 1e9, 1), "frac_of_hbm_peak": round(r_near * alg / 1e9 / HBM_PEAK_GBS, 4),
                    "cost_first": c_near, "kernels": per, "roofline": roof})
    if "start" in legs:
        r_start, c_start, per, roof = rate(start, vol, "start")
        out.update({"evals_per_sec_at_untilted_start": round(r_start, 1), "cost_first_at_start": c_start,
                    "untilted_start": {"evals_per_sec": round(r_start, 1), "kernels": per, "roofline": roof}})
    if "dense" in legs:
        # the same evaluation on a volume without zero voxels (the gradient kernels clip every ray to the bounding box of the
        # non-zero voxels; Shepp-Logan fills about half of its cube)
        dense = be.empty(N ** 3)
        be.fill(dense, 0.05)
        be.axpy(dense, vol, 1.0)
        r_dense, _, per, roof = rate(near, dense, "dense")
        out["dense_volume"] = {"evals_per_sec": round(r_dense, 1), "frac_of_hbm_peak": round(r_dense * alg / 1e9 / HBM_PEAK_GBS, 4),
                               "config": "same poses and measured projections, volume = Shepp-Logan + 0.05 (no zero voxel)", "kernels": per, "roofline": roof}
    return out


def grad_roofline(per, alg_bytes_per_pass, key, samples_per_pass):
    """The gradient kernels of one evaluation pass as ONE unit of work (a pass is one launch per variant in use): summed live time against
    the summed counters of those launches.  Units: HBM (FETCH/WRITE counters), VALU busy, SALU issue (one scalar instruction per clock per
    CU), TA busy (the texture-address path the gathers go through; profiles/round3_grad_counters.md) -- `bound` = the busiest."""
    if not per:
        return None
    t = sum(v["ms_per_pass"] for v in per.values()) * 1e-3
    alg_gbs = alg_bytes_per_pass / t / 1e9
    r = {"kernels": sorted(per), "ms_per_pass": round(t * 1e3, 3), "hbm_algorithmic_frac": round(alg_gbs / HBM_PEAK_GBS, 4), "hbm_counter_frac": None,
         "traffic": None, "counters": None,
         "useful_flop_frac": round(30.0 * samples_per_pass / t / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
         "useful_flop_note": "30 flop per in-volume sample: value 7 lerps x 2 + three gradient components (4 differences + 3 lerps x 2 each, shared partly): "
                             "a count of the trilinear value + gradient arithmetic, not of the instructions spent"}
    stale, sq, pmc, srcs = [], {}, 0.0, []
    have_sq = have_pmc = True
    for name in per:
        ent, src = load_counters("sq_counters.json", key, name, stale)
        if ent is None:
            have_sq = False
        else:
            srcs.append(src)
            for k, v in ent.items():
                if isinstance(v, (int, float)):
                    sq[k] = sq.get(k, 0.0) + v
        ent, src = load_counters("pmc_traffic.json", key, name, stale)
        if ent is None:
            have_pmc = False
        else:
            pmc += ent["hbm_bytes_per_launch"]
    if stale:
        r["stale_counters_refused"] = sorted(set(stale))
    util = {}
    if have_pmc and pmc > 0:
        r["traffic"] = pmc
        r["hbm_counter_frac"] = round(pmc / t / 1e9 / HBM_PEAK_GBS, 4)
        util["hbm"] = (pmc / t / 1e9, HBM_PEAK_GBS, "GB/s")
    if have_sq and sq:
        r["counters"] = {"source": srcs[0] if srcs else None, "per_pass": sq}
        if sq.get("SQ_ACTIVE_INST_VALU"):
            util["valu"] = (4.0 * sq["SQ_ACTIVE_INST_VALU"] / t / 1e9, N_CU * 4 * CLK_GHZ, "G SIMD-cycles/s")
        if sq.get("SQ_INSTS_SALU"):
            util["salu"] = (sq["SQ_INSTS_SALU"] / t / 1e9, N_CU * CLK_GHZ, "G scalar instr/s (one per clock per CU)")
        if sq.get("TA_TA_BUSY_sum"):
            util["ta"] = (sq["TA_TA_BUSY_sum"] / t / 1e9, N_CU * CLK_GHZ, "G TA-busy cycles/s (one TA per CU)")
        if sq.get("SQ_INSTS_VALU"):
            r["valu_instr_per_sample_wave"] = round(sq["SQ_INSTS_VALU"] / (samples_per_pass / 64.0), 2)
        if sq.get("SQ_INSTS_VMEM_RD"):
            r["vmem_loads_per_sample_wave"] = round(sq["SQ_INSTS_VMEM_RD"] / (samples_per_pass / 64.0), 2)
    if not util:
        r.update({"bound": "hbm", "achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / HBM_PEAK_GBS, 4),
                  "note": "no committed PMC counters for workload key %s on these kernel sources: algorithmic-HBM figure" % key})
        return r
    bound = max(util, key=lambda k: util[k][0] / util[k][1])
    ach, peak, unit = util[bound]
    r.update({"bound": bound, "achieved": round(ach, 1), "peak": round(peak, 1), "unit": unit, "frac": round(ach / peak, 4),
              "utilisation": {k: {"achieved": round(v[0], 1), "peak": round(v[1], 1), "unit": v[2], "frac": round(v[0] / v[1], 4)} for k, v in util.items()}})
    return r


def align_rigid_e2e(comm, ctx, rank, world, sharded, N=512, n_proj=720, sirt_iters=10, n_outer=2):
    """Side measurement (not `value`): BASELINE config 5 END TO END -- `n_outer` outer iterations of the reference's examples/align_rigid.py:27-59
    loop on N^3 x n_proj with +-2 deg / +-5 px pose errors (default_rng(5), as align_rate): `sirt_iters` SIRT iterations with positivity
    (device-resident solver, warm-started in HBM), then ONE lock-step alignment pass (every projection's scipy L-BFGS-B on cost_xzab /
    gradient_xzab from a zero start, bounds +-6 px / +-0.05 rad, one fused cost+gradient launch per round of evaluations).
    The FIRST outer iteration reconstructs at the nominal -- flat -- poses; from the second on the poses are the recovered, tilted ones and
    SIRT runs on the general tile kernels: `outer` carries each iteration's wall and kernel times.
    `sharded` (world > 1, or --force-sharded on a 1-rank communicator): tomography_alignment_amd.examples.align_rigid.run(comm=...) -- the
    angle-sharded SIRT and every rank aligning its own np.array_split block of the projections; each rank generates, keeps and uses only
    its own measured rows.  Wall times are the maximum over the ranks and include the host side (scipy's L-BFGS-B steps, staging);
    kernel times are rank 0's HIP-event sums.  The last alignment pass is priced against a REPLAY of its own evaluations in full batches
    (`end_to_end_over_full_batch_replay`, 1 = nothing lost to batching or to the host)."""
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.examples import align_rigid
    from tomography_alignment_amd.recon import sirt_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry
    from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN
    rng = np.random.default_rng(5)
    phi = np.linspace(0., np.pi, n_proj)
    alpha, beta = np.deg2rad(rng.uniform(-2, 2, n_proj)), np.deg2rad(rng.uniform(-2, 2, n_proj))
    xyz = np.zeros((n_proj, 3))
    xyz[:, 0], xyz[:, 2] = rng.uniform(-5, 5, n_proj), rng.uniform(-5, 5, n_proj)
    geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    mine = np.array_split(np.arange(n_proj), world)[rank]
    be = HipBackend(sirt_mpi.SIRT._shard_geometry(geo, mine), ctx=ctx)
    vol = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
    d_b = be.forward(_lib.poses_array(phi[mine], alpha[mine], beta[mine], xyz[mine], np.zeros(3)), vol, be.empty(mine.size * N * N))
    data = dict(projections=d_b, projections_shape=(n_proj, N, N), ny=N, phi=phi, alpha=alpha, beta=beta, xyz=xyz, phantom=vol)
    names = ("k_fwd_tile", "k_fwd_tile_flat", "k_adj_tile", "k_adj_tile_flat", "k_adj_gather_flat", "k_cost_grad", "k_pad", "k_box", "k_residual_scale", "k_update", "k_vec",
             "k_absmax", "allreduce_f32", "reduce_scatter_f32", "allgather_f32")
    ctx.sync()
    comm.barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    trace = []
    _, a_rec, b_rec, xyz_rec, hist, loop = align_rigid.run(data, n_outer=n_outer, sirt_iters=sirt_iters, bounds=((-6., 6.), (-6., 6.), (-0.05, 0.05), (-0.05, 0.05)),
                                                           verbose=False, backend=be, align_kwargs={"trace": trace}, comm=comm if sharded else None,
                                                           kernel_names=names, download=False, return_loop=True)
    ctx.sync()
    comm.barrier()
    wall = comm.allreduce_max(time.perf_counter() - t0)
    ctx.profile_enable(False)
    # The kernel rate the LAST pass could have had: the very evaluations it made (same volume -- the pass ran against the reconstruction
    # still in HBM --, same poses, same measured rows), replayed in launches of a rank's whole block.  alignment_gradient.evals_per_sec is
    # taken on another pose population (all 0.5 deg from the truth); the optimisers' own points run from untilted starts to tilts on
    # the bounds, so only this replay prices the loop's batching + host side.
    last = trace[len(trace) - int(hist[-1]["launches"]):]
    rows_all = np.concatenate([t[0] for t in last]) if last else np.zeros(0, np.int64)
    poses_all = np.concatenate([t[1] for t in last]) if last else np.zeros((0, _lib.POSE_STRIDE))
    replay_s, m = 0.0, max(1, mine.size)
    if rows_all.size:
        be.cost_grad(np.ascontiguousarray(poses_all[:m]), loop.d_rec, loop.d_b, rows=rows_all[:m])
        ctx.set_option("reuse_staged_volume", 1)
        ctx.sync()
        t0 = time.perf_counter()
        for a in range(0, rows_all.size, m):
            be.cost_grad(np.ascontiguousarray(poses_all[a:a + m]), loop.d_rec, loop.d_b, rows=rows_all[a:a + m])
        ctx.sync()
        replay_s = time.perf_counter() - t0
        ctx.set_option("reuse_staged_volume", 0)
    replay_s = comm.allreduce_max(replay_s)
    outer = []
    for h in hist:
        e = {"outer": h["outer"], "sirt_wall_s": comm.allreduce_max(h["sirt_wall_s"]), "align_wall_s": comm.allreduce_max(h["align_wall_s"]),
             "sirt_iterations": h["sirt_iterations"], "rmse_after_sirt": h["rmse"], "residual_after_alignment": h["residual"],
             "alignment_evals_this_rank": int(h["evals"]), "alignment_launches_this_rank": int(h["launches"]),
             "alignment_evals": int(round(comm.allreduce_scalar(float(h["evals"])))),
             "shift_err_px": h["shift_err_px"], "tilt_err_deg": h["tilt_err_deg"],
             "sirt_kernel_ms": h.get("sirt_kernel_ms", {}), "align_kernel_ms": h.get("align_kernel_ms", {})}
        e["sirt_ms_per_iteration"] = round(1e3 * e["sirt_wall_s"] / max(1, h["sirt_iterations"]), 2)
        e["evals_per_sec_end_to_end"] = round(e["alignment_evals"] / max(1e-9, e["align_wall_s"]), 1)
        outer.append(e)
    hl = outer[-1]
    return {"wall_s": round(wall, 2), "unit": "s", "ranks": world, "sharded_code_path": bool(sharded),
            "config": "%d^3 volume, %d projections, +-2 deg / +-5 px pose errors: %d outer iterations of [%d SIRT iterations (positivity; outer 0 at the nominal poses, "
                      "then at the recovered ones) + one lock-step L-BFGS-B alignment pass (tx, tz, alpha, beta from zero, bounds +-6 px / +-0.05 rad)]%s"
                      % (N, n_proj, n_outer, sirt_iters, "; angles sharded over %d rank(s): sirt_mpi.SIRT + align_projections_sharded" % world if sharded else ""),
            "outer": outer,
            "sirt_wall_s_flat_poses": outer[0]["sirt_wall_s"], "sirt_wall_s_recovered_poses": outer[-1]["sirt_wall_s"] if n_outer > 1 else None,
            "align_wall_s": hl["align_wall_s"], "alignment_evals": hl["alignment_evals"],
            "evals_per_sec_end_to_end": hl["evals_per_sec_end_to_end"],
            "evals_per_sec_same_evaluations_in_full_batches": round(comm.allreduce_scalar(float(rows_all.size)) / max(1e-9, replay_s), 1),
            "end_to_end_over_full_batch_replay": round(replay_s / max(1e-9, hl["align_wall_s"]), 3),
            "driver": hist[-1].get("driver"),
            "shift_err_px": {"before": float(np.abs(xyz[:, [0, 2]]).mean()), "after": [o["shift_err_px"] for o in outer]},
            "tilt_err_deg": {"before": float(np.rad2deg(np.abs(np.column_stack([alpha, beta])).mean())), "after": [o["tilt_err_deg"] for o in outer]}}


def _cpu_share():
    """Cores this process may actually use: the cgroup CPU quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(be, d_true, N, n_proj, phi):
    """Host-CPU baselines on this box, each on a bounded sample of the same workloads:
      value / all_cores   the CPU oracle (oracle/: plain-C port of the reference algorithm, 2-3x FASTER than the reference's own
                          Fortran on the same core), forward + exact adjoint of the N^3 workload:
                          ONE thread on every 8th detector row of 2 angles; OpenMP over rays on the box's CPU share on 16 whole
                          angles (everything in full when N <= 256: configs 1 and 2) -- SIRT iterations/s, extrapolated linearly;
      gradient            the oracle's projection + 6-DoF gradient (src/ray_wt_grad.f90:95-223 restated) of config-5 poses:
                          evals/s on one thread (1/8 of a pose's rays) and on all cores (two whole poses);
      reference           the REFERENCE ITSELF: its untouched Fortran (forward_project_, compute_gradient_) built by
                          oracle/build_ref.sh into oracle/_ref/libref_mf.so, timed by tools/ref_baseline.py in a child process
                          (kind "reference"); its SIRT back-projection is a scipy product of a matrix that cannot exist at this
                          size, so the exact adjoint stays a port figure."""
    from oracle import oracle as orc
    x = be.download(d_true)
    n_cpu = _cpu_share()
    full = N <= 256
    orc.set_threads(1)
    og1 = orc.Geo(1, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    rows = np.arange(N) if full and N <= 128 else np.arange(3, N, 8)
    ray_idx = (rows[:, None] * N + np.arange(N)[None, :]).ravel()
    frac = ray_idx.size / float(N * N)
    picks = [n_proj // 3, (2 * n_proj) // 3 + 1]             # generic (non axis-aligned) angles
    tf = ta = 0.0
    for ip in picks:
        t0 = time.perf_counter()
        ax = orc.forward_rays(og1, x, phi[ip], ray_idx)
        t1 = time.perf_counter()
        orc.adjoint_rays(og1, ax.astype(np.float32), phi[ip], ray_idx)
        t2 = time.perf_counter()
        tf += (t1 - t0) / frac / len(picks)
        ta += (t2 - t1) / frac / len(picks)
    out = {"value": round(1.0 / ((tf + ta) * n_proj), 8), "unit": "it/s", "cores": 1, "kind": "port",
           "sample": "%s detector rows of %d of %d angles of the same %d^3 workload on one host core (forward %.2f s + exact adjoint %.2f s per "
                     "whole angle), extrapolated linearly in rays and angles" % ("all" if rows.size == N else "every 8th of the", len(picks), n_proj, N, tf, ta),
           "forward_s": round(tf * n_proj, 3), "backproj_s": round(ta * n_proj, 3),
           "host_cpus": os.cpu_count(), "cpu_share": n_cpu,
           "calibration": "the oracle runs 2-3x faster than the reference's own flang-built Fortran on one core (BASELINE.md section 4; `reference` below, same run)"}
    if n_cpu > 1:
        n_ang = n_proj if full else (16 if n_cpu >= 8 else 2)
        orc.set_threads(n_cpu)
        og = orc.Geo(n_ang, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
        ph = phi if full else phi[n_proj // 3: n_proj // 3 + n_ang]
        t0 = time.perf_counter()
        ax = orc.forward(og, x, phi=ph)
        t1 = time.perf_counter()
        orc.adjoint(og, ax.astype(np.float32), phi=ph, coloured_rows=True)     # threads own whole detector rows: no atomics
        t2 = time.perf_counter()
        fa, aa = (t1 - t0) / n_ang, (t2 - t1) / n_ang
        out["all_cores"] = {"value": round(1.0 / ((fa + aa) * n_proj), 8), "unit": "it/s", "cores": n_cpu,
                            "forward_s": round(fa * n_proj, 3), "backproj_s": round(aa * n_proj, 3),
                            "sample": "%s (forward %.3f s + adjoint %.3f s per angle on %d threads, OpenMP over rays)"
                                      % ("all %d angles, in full" % n_proj if full else "%d of %d whole angles" % (n_ang, n_proj), fa, aa, n_cpu)}
    # ---- alignment gradient (config 5; the size align_rate() uses for this N)
    Ng = min(512, max(32, N // 2))
    rng = np.random.default_rng(5)
    n5 = 720
    phi5 = np.linspace(0., np.pi, n5)
    alpha5, beta5 = np.deg2rad(rng.uniform(-2, 2, n5)), np.deg2rad(rng.uniform(-2, 2, n5))
    tx5, tz5 = rng.uniform(-5, 5, n5), rng.uniform(-5, 5, n5)
    from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN
    xg32 = be.download(be.phantom(be.empty(Ng ** 3), (Ng, Ng, Ng), SHEPP_LOGAN))
    xg = np.ascontiguousarray(xg32, np.float64)
    ogg = orc.Geo(1, np.array([Ng, Ng, Ng]), np.ones(3), np.array([Ng, Ng]), np.ones(2))
    pose = lambda i: (alpha5[i], beta5[i], phi5[i], np.array([tx5[i], 0.0, tz5[i]]), np.zeros(3))      # noqa: E731
    orc.set_threads(1)
    rows = np.arange(3, Ng, 8)
    idx = (rows[:, None] * Ng + np.arange(Ng)[None, :]).ravel()
    t0 = time.perf_counter()
    orc.projection_gradient_rays(ogg, xg, *pose(240), idx)
    tg1 = (time.perf_counter() - t0) * (Ng * Ng) / idx.size
    grad = {"evals_per_sec": round(1.0 / tg1, 5), "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "every 8th detector row of one of the 720 config-5 poses (%d^3, +-2 deg / +-5 px) on one host core: %.2f s per whole evaluation, "
                      "extrapolated linearly in rays" % (Ng, tg1)}
    if n_cpu > 1:
        orc.set_threads(n_cpu)
        all_idx = np.arange(Ng * Ng)
        t0 = time.perf_counter()
        for i in (240, 481):
            orc.projection_gradient_rays(ogg, xg, *pose(i), all_idx)
        tga = (time.perf_counter() - t0) / 2
        grad["all_cores"] = {"evals_per_sec": round(1.0 / tga, 4), "cores": n_cpu, "sample": "two whole config-5 poses on %d threads (OpenMP over rays): %.2f s per evaluation" % (n_cpu, tga)}
    orc.set_threads(1)
    out["gradient"] = grad
    del xg
    # ---- the reference itself (a compiled binary of its untouched Fortran), in a child process
    out["reference"] = reference_baseline(x, N, n_proj, xg32, Ng)
    return out


def reference_baseline(x, N, n_proj, xg32, Ng):
    """cpu_baseline.reference: tools/ref_baseline.py on volumes handed over through /dev/shm (or the temp dir)."""
    import tempfile
    lib = os.path.join(ROOT, "oracle", "_ref", "libref_mf.so")
    if not os.path.exists(lib):
        return {"kind": "reference", "available": False, "reason": "oracle/_ref/libref_mf.so is not in this tree (built by oracle/build_ref.sh where /root/reference exists)"}
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    f1, f2 = os.path.join(d, "tomo_bench_%d_fwd.npy" % os.getpid()), os.path.join(d, "tomo_bench_%d_grad.npy" % os.getpid())
    try:
        np.save(f1, np.ascontiguousarray(x, np.float32).ravel())
        np.save(f2, np.ascontiguousarray(xg32, np.float32).ravel())
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_baseline.py"), lib, f1, str(N), str(n_proj), f2, str(Ng)],
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0 or not r.stdout.strip():
            return {"kind": "reference", "available": False, "reason": "tools/ref_baseline.py exited with %s: %s" % (r.returncode, (r.stderr or "").strip()[-300:])}
        j = json.loads(r.stdout.strip().splitlines()[-1])
        j["available"] = True
        return j
    except (OSError, ValueError, subprocess.TimeoutExpired) as e:
        return {"kind": "reference", "available": False, "reason": repr(e)[:300]}
    finally:
        for f in (f1, f2):
            try:
                os.remove(f)
            except OSError:
                pass


def copy_probe(ctx, be, n_bytes=1 << 31):
    """Measured device-to-device copy rate (read + write bytes per second) -- the practical ceiling next to the 8 TB/s spec."""
    n = n_bytes // 4
    a, b = be.zeros(n), be.empty(n)
    b.copy_from(a)
    ctx.sync()
    ctx.timer_start()
    for _ in range(4):
        b.copy_from(a)
    ms = ctx.timer_stop()
    return round(4 * 2.0 * n_bytes / (ms * 1e-3) / 1e9, 1)


if __name__ == "__main__":
    main()
