#!/usr/bin/env python3
"""
bench.py -- headline benchmark of BASELINE.json: SIRT iterations/s on a 1024^3 volume x 1024 angles
(parallel beam, Shepp-Logan, phi = linspace(0, pi)), with the forward / back-projection kernels priced
against the MI355X HBM roofline.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment; torch is not imported).
A "step" is one SIRT iteration: A.rec, residual, A^T(W*res), all-reduce of the voxel update over the
angle shards (RCCL over xGMI; strong scaling: the 1024 angles are split across the N GPUs), update.
Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Extra objects in the line ("Measurement"):
  roofline     dominant kernel of the step: algorithmic bytes per launch / mean launch time (HIP events
               recorded on the kernel's own stream inside the timed region) against 8 TB/s
  cpu_baseline the CPU oracle (oracle/: plain-C port of the reference algorithm, 1 thread) timed on this
               box's host cores on ONE angle of the same workload, extrapolated linearly in n_proj
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def main():
    # Libraries (RCCL prints a version banner at communicator init) must not pollute the ONE JSON line on stdout:
    # send everything written to fd 1 during the run to stderr and keep the real stdout for the final line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="volume edge N (N^3 voxels, N x N detector)")
    ap.add_argument("--angles", type=int, default=1024)
    ap.add_argument("--perturbed", action="store_true", help="alpha,beta ~ U(+-1 deg), tx,tz ~ U(+-2 px) (default_rng(0))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-align", action="store_true", help="skip the alignment-gradient evals/s side measurement (config 5)")
    ap.add_argument("--no-tilted", action="store_true", help="skip the tilted-pose SIRT side measurement")
    ap.add_argument("--force-sharded", action="store_true",
                    help="N=1 only: run the multi-GPU code path (sharded solver, x-slab pipelined all-reduce) on a 1-rank RCCL communicator")
    ap.add_argument("--fwd-variant", type=int, default=None)
    ap.add_argument("--adj-variant", type=int, default=None)
    args = ap.parse_args()

    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.backend import HipBackend
    from tomography_alignment_amd.comm import RcclComm
    from tomography_alignment_amd.recon import sirt as sirt_mod, sirt_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry
    from tomography_alignment_amd.utilities.generate_phantom import SHEPP_LOGAN

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
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

    geo = Geometry(n_proj, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
    phi = np.linspace(0., np.pi, n_proj)
    alpha, beta, xyz = np.zeros(n_proj), np.zeros(n_proj), np.zeros((n_proj, 3))
    if args.perturbed:
        rng = np.random.default_rng(0)
        alpha = np.deg2rad(rng.uniform(-1, 1, n_proj))
        beta = np.deg2rad(rng.uniform(-1, 1, n_proj))
        xyz[:, 0] = rng.uniform(-2, 2, n_proj)
        xyz[:, 2] = rng.uniform(-2, 2, n_proj)
    angles = np.array([phi, alpha, beta]).T

    # ---- synthetic data, generated and kept on the device: phantom -> this rank's sinogram rows
    my_rows = np.array_split(np.arange(n_proj), world)[rank]
    shard_geo = sirt_mpi.SIRT._shard_geometry(geo, my_rows)
    be = HipBackend(shard_geo, ctx=ctx)
    d_true = be.phantom(be.empty(N ** 3), (N, N, N), SHEPP_LOGAN)
    poses = _lib.poses_array(phi[my_rows], alpha[my_rows], beta[my_rows], xyz[my_rows], np.zeros(3))
    d_b = be.forward(poses, d_true, be.empty(my_rows.size * N * N))
    opts = {"_backend": be}
    if world > 1 or args.force_sharded:
        solver = sirt_mpi.SIRT(comm, geo, d_b, angles, xyz, opts)
    else:
        solver = sirt_mod.SIRT(geo, d_b, angles, xyz, opts)

    def barrier():
        ctx.sync()
        comm.barrier()
        ctx.sync()

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

    # ---- per-kernel timing of the timed region (HIP events on the ctx stream)
    kern = {}
    for name in ("k_fwd_v1", "k_fwd_v2", "k_fwd_tile", "k_fwd_tile_flat", "k_adj_v1", "k_adj_tile", "k_adj_tile_flat", "k_adj_gather_flat", "k_pad", "k_unpad", "k_absmax", "k_residual_scale", "k_update",
                 "k_vec", "allreduce_f32"):
        n, ms = ctx.profile_get(name)
        if n:
            # a pass over all angles may be issued as several launches (x slabs of the pipelined back-projection):
            # ms_per_step sums them, so bytes-per-pass / ms_per_step == bytes-per-launch / avg launch time
            kern[name] = {"launches": n, "avg_ms": ms / n, "launches_per_step": n / float(args.steps), "ms_per_step": ms / float(args.steps)}
    n_loc = my_rows.size
    n_det = N * N
    alg_fwd = n_loc * (4.0 * N ** 3 + 4.0 * n_det)               # bytes per forward launch   
    alg_adj = n_loc * (8.0 * N ** 3 + 4.0 * n_det)               # bytes per back-projection launch
    fwd_name = next((k for k in ("k_fwd_tile_flat", "k_fwd_tile", "k_fwd_v2", "k_fwd_v1") if k in kern), None)
    adj_name = next((k for k in ("k_adj_gather_flat", "k_adj_tile_flat", "k_adj_tile", "k_adj_v1") if k in kern), None)
    cands = []
    if fwd_name:
        cands.append((kern[fwd_name]["ms_per_step"], fwd_name, alg_fwd))
    if adj_name:
        cands.append((kern[adj_name]["ms_per_step"], adj_name, alg_adj))
    roofline = None
    if cands:
        step_ms, name, alg = max(cands)
        lps = kern[name]["launches_per_step"]
        ach = alg / (step_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": alg / lps, "avg_launch_ms": round(step_ms / lps, 3), "launches_per_step": lps}
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (counters cannot be
        # collected from inside the timed run; see profiles/*_rocprof_summary.md for how they were taken/corrected)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pmc.get("key") == "N%d_A%d_G%d%s" % (N, n_proj, world, "_P" if args.perturbed else "") and name in pmc["kernels"]:
                roofline["traffic"] = pmc["kernels"][name]["hbm_bytes_per_launch"] / lps
                roofline["traffic_source"] = "profiles/pmc_traffic.json (%s)" % pmc.get("source", "")
        except (OSError, ValueError, KeyError):
            pass
    extra = {}
    if fwd_name:
        extra["forward_alg_GBps"] = round(alg_fwd / (kern[fwd_name]["ms_per_step"] * 1e-3) / 1e9, 1)
    if adj_name:
        extra["backproj_alg_GBps"] = round(alg_adj / (kern[adj_name]["ms_per_step"] * 1e-3) / 1e9, 1)

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
        "data": "synthetic (3-D Shepp-Logan generated on the device; sinogram = its forward projection)",
        "config": {"workload": "SIRT %d^3 volume x %d angles, parallel beam, step 1.0, detector %dx%d%s"
                               % (N, n_proj, N, N, ", perturbed poses" if args.perturbed else ""),
                   "sharding": "angles split over %d GPU(s), RCCL all-reduce of the voxel update" % world,
                   "rms_error_last": float(rms[-1])},
        "roofline": roofline,
        "kernels": kern,
    }
    out.update(extra)
    if roofline is not None and fwd_name and adj_name:
        step_alg = n_proj * (12.0 * N ** 3 + 8.0 * n_det) + 16.0 * n_proj * n_det + 16.0 * N ** 3
        out["sirt_step_alg_GBps"] = round(step_alg / (elapsed / args.steps) / 1e9, 1)

    if not args.no_tilted and not args.perturbed:
        # side measurement (not `value`): the same workload with tilted poses (alpha, beta ~ U(+-1 deg), tx, tz ~ U(+-2 px),
        # default_rng(0) -- SURVEY 8d's perturbed run): these take the general tile kernels, which is what SIRT runs on
        # once an alignment pass has moved the poses
        del solver
        rng = np.random.default_rng(0)
        alpha_t, beta_t = np.deg2rad(rng.uniform(-1, 1, n_proj)), np.deg2rad(rng.uniform(-1, 1, n_proj))
        xyz_t = np.zeros((n_proj, 3))
        xyz_t[:, 0], xyz_t[:, 2] = rng.uniform(-2, 2, n_proj), rng.uniform(-2, 2, n_proj)
        angles_t = np.array([phi, alpha_t, beta_t]).T
        poses_t = _lib.poses_array(phi[my_rows], alpha_t[my_rows], beta_t[my_rows], xyz_t[my_rows], np.zeros(3))
        be.forward(poses_t, d_true, d_b)
        if world > 1 or args.force_sharded:
            solver = sirt_mpi.SIRT(comm, geo, d_b, angles_t, xyz_t, opts)
        else:
            solver = sirt_mod.SIRT(geo, d_b, angles_t, xyz_t, opts)
        solver.iterate_device(niter=1)
        barrier()
        t0 = time.perf_counter()
        solver.iterate_device(niter=2)
        barrier()
        dt = comm.allreduce_max(time.perf_counter() - t0)
        out["tilted_poses"] = {"value": round(2.0 / dt, 6), "unit": "it/s", "steps": 2, "warmup": 1,
                               "config": "same workload, alpha, beta ~ U(+-1 deg), tx, tz ~ U(+-2 px): general tile kernels"}
    if not args.no_align:
        del solver
        out["alignment_gradient"] = align_rate(comm, ctx, rank, world, N=min(512, max(32, N // 2)), n_proj=720 if N >= 1024 else max(8, n_proj // 2))
    if roofline is not None:
        roofline["measured_d2d_copy_GBps"] = copy_probe(ctx, be)        # read + write of a 2 GiB hipMemcpy D2D, same run
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(be, d_true, N, n_proj, phi)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_sharded:
        comm.close()


def align_rate(comm, ctx, rank, world, N=512, n_proj=720, passes=3):
    """Side measurement (not `value`): alignment cost+gradient evaluations per second on BASELINE config 5 -- 512^3
    volume, 720 projections with +-2 deg / +-5 px perturbations (default_rng(5)); projections are sharded over the
    ranks with a replicated volume and no collective; one fused launch evaluates a rank's whole shard."""
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
    rates = {}
    for tag, poses in (("near_truth", near), ("start", start)):
        be.cost_grad(poses, vol, b)
        ctx.sync()
        comm.barrier()
        t0 = time.perf_counter()
        for _ in range(passes):
            cost, g6 = be.cost_grad(poses, vol, b)
        ctx.sync()
        comm.barrier()
        dt = comm.allreduce_max(time.perf_counter() - t0)
        rates[tag] = (passes * n_proj / dt, float(cost[0]))
    rate = rates["near_truth"][0]
    alg = 4.0 * N ** 3 + 4.0 * N * N + 28.0                      # fused form, BASELINE.md section 3
    return {"evals_per_sec": round(rate, 1), "unit": "evals/s", "config": "%d^3 volume, %d projections simulated with +-2 deg / +-5 px pose "
            "errors, fused cost+6-DoF gradient evaluated 0.5 deg / 1 px away from the true poses" % (N, n_proj),
            "alg_GBps": round(rate * alg / 1e9, 1), "frac_of_hbm_peak": round(rate * alg / 1e9 / HBM_PEAK_GBS, 4),
            "evals_per_sec_at_untilted_start": round(rates["start"][0], 1),
            "projections_per_launch": int(mine.size), "cost_first": rates["near_truth"][1], "cost_first_at_start": rates["start"][1]}


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
    """Time the CPU oracle (plain-C port of the reference algorithm) on a bounded sample of the same workload, forward +
    adjoint, extrapolated linearly in n_proj: (i) ONE thread, like the reference's serial Fortran, on one projection angle;
    (ii) all host cores (OpenMP over rays) on a few angles."""
    from oracle import oracle as orc
    x = be.download(d_true)
    lib = orc._lib()
    n_cpu = _cpu_share()

    def one(n_ang, threads):
        lib.orc_set_threads(int(threads))
        og = orc.Geo(n_ang, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
        ph = phi[n_proj // 3: n_proj // 3 + n_ang]           # generic (non axis-aligned) angles
        t0 = time.perf_counter()
        ax = orc.forward(og, x, phi=ph)
        t1 = time.perf_counter()
        orc.adjoint(og, ax.astype(np.float32), phi=ph, coloured_rows=threads > 1)     # threads own whole detector rows: no atomics
        t2 = time.perf_counter()
        return (t1 - t0) / n_ang, (t2 - t1) / n_ang

    f1, a1 = one(1, 1)
    out = {"value": round(1.0 / ((f1 + a1) * n_proj), 8), "unit": "it/s", "cores": 1, "kind": "port",
           "sample": "1 of %d angles of the same %d^3 workload (forward %.1f s + adjoint %.1f s on one host core), "
                     "extrapolated linearly in n_proj" % (n_proj, N, f1, a1),
           "host_cpus": os.cpu_count(), "cpu_share": n_cpu}
    if n_cpu > 1:
        n_ang = 2 if n_cpu >= 8 else 1
        fa, aa = one(n_ang, n_cpu)
        out["all_cores"] = {"value": round(1.0 / ((fa + aa) * n_proj), 8), "unit": "it/s", "cores": n_cpu,
                            "sample": "%d of %d angles (forward %.2f s + adjoint %.2f s per angle on %d threads, OpenMP over rays)"
                                      % (n_ang, n_proj, fa, aa, n_cpu)}
    lib.orc_set_threads(1)
    return out


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
