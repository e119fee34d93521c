"""The N>1 path on CPU: world_size 1 and 2 over torch.distributed/gloo.  The angle-sharded solvers must give the
unsharded answer (sum over shards == whole) with exactly one volume all-reduce per SIRT iteration."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_max


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, out, worker="_gloo_worker.py"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=600)[0].decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                      # exactly the processes started here
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return np.load(out)


@pytest.mark.timeout(900)
def test_sharded_solvers_match_unsharded(tmp_path):
    one = _run(1, str(tmp_path / "w1.npz"))
    two = _run(2, str(tmp_path / "w2.npz"))
    assert rel_max(two["rec"], one["rec"]) < 1e-5           # float32 sums in a different order, nothing else
    assert np.allclose(two["err"], one["err"], rtol=1e-5)
    assert rel_max(two["crec"], one["crec"]) < 1e-5
    assert np.allclose(two["cerr"], one["cerr"], rtol=1e-5)
    # CGLS (round 5): world 2 pipelines -- per iteration one reduce-scatter and one all-gather per slab (3 at 32^3), no whole-volume
    # all-reduce after the constructor's (recon/cgls_mpi.py:55), iterations 2.. make A p slab by slab behind the all-gathers (forward calls:
    # 1 at init + 1 whole + 3 x 2 slab calls for 3 iterations made ahead; none is made ahead after the last); world 1 is the plain sequence
    n_ci = len(two["cerr"])
    assert list(two["c_counts"][:5]) == [3 * n_ci, 3 * n_ci, 0, 1, 1], two["c_counts"]
    assert list(one["c_counts"][:5]) == [0, 0, 0, 1 + n_ci, 0], one["c_counts"]
    assert int(one["c_counts"][5]) == 1 + n_ci and int(two["c_counts"][5]) == 2 + 2 * (n_ci - 1), (one["c_counts"], two["c_counts"])
    for w in (one, two):
        for tag in ("forced", "allreduce", "plain"):      # slab pipeline forced (also at world 1) / all-reduce form / plain sequence
            assert rel_max(w["c_%s_rec" % tag], one["crec"]) < 1e-5 and np.allclose(w["c_%s_err" % tag], one["cerr"], rtol=1e-5), tag
        assert rel_max(w["c_gt_rec"], one["c_gt_rec"]) < 1e-5 and np.allclose(w["c_gt_err"], one["c_gt_err"], rtol=1e-5)
    # world 2 pipelines (decided collectively at init): one whole-volume all-reduce of V at init (recon/sirt_mpi.py:68), then per
    # iteration one all-reduce per x slab (32^3: 3 tile columns -> 3 slabs) whose sizes add up to the volume (:103); the forward
    # projection of iterations 2.. is made slab by slab behind the update (2 whole forwards: W at init and iteration 1)
    n_it = len(two["err"])
    assert bool(two["pipelined"]) and not bool(one["pipelined"])
    # round 4: a slab's sum is a reduce-scatter (each rank receives -- and updates -- its own half; the stand-in communicator fills the
    # other half with NaN, so a rank that read it would show) followed by an all-gather of the updated pieces; no slab all-reduce
    # is left (every slab of this volume splits evenly over 2 ranks), and every all-gather is waited for exactly once
    assert int(two["n_allreduce_sirt"]) == 1 and int(two["n_slab_sirt"]) == 0
    assert int(two["n_rs"]) == 3 * n_it and int(two["n_ag"]) == 3 * n_it and int(two["n_wg"]) == 3 * n_it
    assert int(two["slab_sizes"].sum()) == 32 ** 3
    assert int(one["n_allreduce_sirt"]) == 1 + len(one["err"]) and int(one["n_slab_sirt"]) == 0 and int(one["n_rs"]) == 0
    # the all-reduce form of the pipeline (shard_update = False: round 3) gives the same reconstruction with 3 slab all-reduces per iteration
    assert int(two["n_slab_allreduce_form"]) == 3 * n_it and rel_max(two["rec_a"], two["rec"]) < 1e-5 and np.allclose(two["err_a"], two["err"], rtol=1e-5)
    # with a ground truth the error curve is the same on EVERY rank in both forms (a rank that saw another curve could stop alone and leave
    # its peers in a collective) and equals the unsharded one
    for w in (one, two):
        assert float(w["rank_spread"]) < 1e-12 and np.allclose(w["err_gt_sharded"], one["err_gt_sharded"], rtol=1e-5) and np.allclose(w["err_gt_allreduce"], one["err_gt_sharded"], rtol=1e-5)
    # world 3: slabs that do not split evenly -- pieces by reduce-scatter, the < 3 left-over voxels of a slab by a small all-reduce
    three = _run(3, str(tmp_path / "w3.npz"))
    assert bool(three["pipelined"]) and int(three["n_rs"]) == 3 * n_it and 0 < int(three["n_slab_sirt"]) <= 3 * n_it
    assert int(three["slab_sizes"].sum()) == 32 ** 3 and int(three["slab_sizes"].min()) < 3
    assert rel_max(three["rec"], one["rec"]) < 1e-5 and np.allclose(three["err"], one["err"], rtol=1e-5)
    assert rel_max(three["rec_g"], one["rec_g"]) < 1e-5 and np.allclose(three["err_g"], one["err_g"], rtol=1e-5)     # error sums over pieces + tails
    for tag in ("forced", "allreduce", "plain", "gt"):      # CGLS with pieces + tails (32^3 slabs do not split evenly over 3 ranks)
        assert rel_max(three["c_%s_rec" % tag], one["c_%s_rec" % tag]) < 1e-5 and np.allclose(three["c_%s_err" % tag], one["c_%s_err" % tag], rtol=1e-5), tag
    assert rel_max(three["crec"], one["crec"]) < 1e-5
    assert float(three["rank_spread"]) < 1e-12 and np.allclose(three["err_gt_sharded"], one["err_gt_sharded"], rtol=1e-5)
    # a rank whose block declines the tile kernels: EVERY rank takes the plain sequence (rank-uniform collectives), same result
    assert not bool(two["declined_pipelined"]) and int(two["declined_n_slab"]) == 0 and int(two["declined_n_vol"]) == 1 + n_it
    assert rel_max(two["rec_d"], two["rec"]) < 1e-5 and np.allclose(two["err_d"], two["err"], rtol=1e-5)
    assert rel_max(one["rec_d"], one["rec"]) < 1e-6
    # pipelined with a ground truth (error accumulated over the slabs) == plain, at world 1 (forced) and world 2
    for w in (one, two):
        assert rel_max(w["rec_g"], w["rec_p"]) < 1e-5 and np.allclose(w["err_g"], w["err_p"], rtol=1e-5)
    assert rel_max(two["rec_g"], one["rec_g"]) < 1e-5
    # Tikhonov gradient descent with scipy's line search: the data terms summed over the ranks == unsharded
    assert rel_max(two["rec_r"], one["rec_r"]) < 1e-5 and np.allclose(two["err_r"], one["err_r"], rtol=1e-5)
    # sharded alignment: every rank aligns its block, the gathered table is the unsharded answer and recovers the poses
    assert np.allclose(two["align_x"], one["align_x"], atol=1e-9) and np.array_equal(two["align_nfev"], one["align_nfev"])
    assert np.allclose(two["align_x"], two["align_true"], atol=2e-4) and np.all(two["align_fun"] < 1e-6)


@pytest.mark.timeout(900)
def test_align_rigid_outer_loop_sharded_matches_unsharded(tmp_path):
    """BASELINE config 5's shape on N ranks: examples/align_rigid.run(comm=...) -- angle-sharded SIRT, then every rank
    aligning its own projections against the replicated reconstruction, twice.  Each HALF of each outer iteration equals the unsharded
    loop's on the same inputs (SIRT: 1e-5, float32 sums in another order; alignment pass: exactly -- same evaluations, same optimiser);
    every rank ends with the same pose table; a rank uploads only its own measured rows; the loop recovers the injected shifts."""
    one = _run(1, str(tmp_path / "e1.npz"))
    two = _run(2, str(tmp_path / "e2.npz"))
    three = _run(3, str(tmp_path / "e3.npz"))
    for w in (one, two, three):
        for stage in (0, 1):
            assert float(w["st_sirt%d_rec" % stage]) < 1e-5 and float(w["st_sirt%d_err" % stage]) < 1e-5, (stage, float(w["st_sirt%d_rec" % stage]))
            assert float(w["st_align%d_x" % stage]) < 1e-12 and float(w["st_align%d_fun" % stage]) < 1e-12 and int(w["st_align%d_nfev" % stage]) == 0
        assert float(w["st_pose_moved"]) > 0.5                      # the second SIRT ran at poses an alignment pass had moved
        assert float(w["e_spread"]) == 0.0                          # every rank holds the same table after the composed loop
        # the composed loops agree as far as the optimiser's own sensitivity allows (a 1e-7 perturbation of the reconstruction moves
        # the UNSHARDED loop's table by up to 0.1 px on this 16^3 problem): first SIRT identical, poses close, same outcome
        assert abs(w["e_rmse"][0] / one["e_rmse"][0] - 1) < 1e-5
        assert np.max(np.abs(w["e_xyz"] - one["e_xyz"])) < 0.2 and abs(w["e_shift_err"][-1] - one["e_shift_err"][-1]) < 0.05
    # a rank's uploads: its own block of the 6 measured rows + the 16 planes of the ground truth, once for both outer iterations
    assert int(one["e_uploaded_rows"]) == 6 + 16 and int(two["e_uploaded_rows"]) == 3 + 16 and int(three["e_uploaded_rows"]) == 2 + 16
    injected = np.abs(one["e_true"][:, :2]).mean()
    assert one["e_shift_err"][-1] < one["e_shift_err"][0] < 0.7 * injected and one["e_rmse"][-1] < one["e_rmse"][0]


@pytest.mark.timeout(900)
def test_world_8_sirt_cgls_and_outer_loop(tmp_path):
    """BASELINE configs 4 / 5 run on 8 GPUs; no box of this pool has more than one.  The bookkeeping an 8-rank run
    depends on, at world 8 over gloo on the CPU (tests/_gloo_worker8.py; world 1 of the same program is the reference):
      * angle blocks of unequal size -- 20 angles as 3 3 3 3 2 2 2 2, 10 projections as 2 2 1 1 1 1 1 1 (config 5's 720 split into 90-pose blocks);
      * the slab plan with 8 slabs, each cut into 8 pieces, slabs 0 and 7 with left-over voxels (1 and 4) that go through the small all-reduce;
      * SIRT and CGLS, pipelined, equal to the one-rank run at 1e-5; the collective counts per iteration; the same counts on every rank;
      * one outer iteration of examples/align_rigid.run(comm=): every rank ends with the same pose table, each half equals the unsharded loop's,
        a rank uploads only its own measured rows."""
    one = _run(1, str(tmp_path / "h1.npz"), "_gloo_worker8.py")
    eight = _run(8, str(tmp_path / "h8.npz"), "_gloo_worker8.py")
    assert list(eight["block_sizes"]) == [3, 3, 3, 3, 2, 2, 2, 2] and list(one["block_sizes"]) == [20]
    assert [np.array_split(np.arange(720), 8)[r].size for r in range(8)] == [90] * 8 and [np.array_split(np.arange(1024), 8)[r].size for r in range(8)] == [128] * 8
    # the slab plan: 8 slabs of 15, 16 x 6 and 12 x planes, together the volume
    px = eight["plan_x"]
    assert px.shape == (8, 2) and px[0, 0] == 0 and px[-1, 1] == 123 and np.array_equal(px[1:, 0], px[:-1, 1]) and list(px[:, 1] - px[:, 0]) == [15] + [16] * 6 + [12]
    n_it = 3
    assert bool(eight["sirt_pipelined"]) and bool(one["sirt_pipelined"]) and bool(eight["cgls_pipelined"])
    # first iteration's collectives in issue order: per slab a reduce-scatter of 8 pieces, then the left-over all-reduce where there is one
    assert list(eight["slab_sizes"][:10]) == [3824, 1, 4080, 4080, 4080, 4080, 4080, 4080, 3056, 4], eight["slab_sizes"]
    # counts [volume all-reduces, small (left-over) all-reduces, reduce-scatters, all-gathers, waits, gather waits]:
    # init: V by one whole-volume all-reduce (recon/sirt_mpi.py:68); per iteration 8 reduce-scatters + 8 all-gathers + 2 left-overs, each waited for once
    assert list(eight["sirt_init_counts"][:4]) == [1, 0, 0, 0]
    assert list(eight["sirt_counts"]) == [0, 2 * n_it, 8 * n_it, 8 * n_it, 10 * n_it, 8 * n_it], eight["sirt_counts"]
    assert list(one["sirt_counts"])[:4] == [0, 0, 8 * n_it, 8 * n_it]                 # one rank: a slab is one piece, nothing left over
    assert list(eight["cgls_counts"][:4]) == [0, 2 * n_it, 8 * n_it, 8 * n_it], eight["cgls_counts"]
    assert rel_max(eight["sirt_rec"], one["sirt_rec"]) < 1e-5 and np.allclose(eight["sirt_err"], one["sirt_err"], rtol=1e-5)
    assert rel_max(eight["cgls_rec"], one["cgls_rec"]) < 1e-5 and np.allclose(eight["cgls_err"], one["cgls_err"], rtol=1e-5)
    assert one["sirt_err"][-1] < one["sirt_err"][0] and one["cgls_err"][-1] < one["cgls_err"][0]
    assert not np.any(eight["counts_spread"]), eight["counts_spread"]                 # every rank issued the same collectives
    # the outer loop: same table everywhere, halves equal to the unsharded loop's, uploads = own rows (2 for rank 0) + the 16 ground-truth planes
    for w in (one, eight):
        assert float(w["e_spread"]) == 0.0 and float(w["st_sirt_rec"]) < 1e-5 and float(w["st_sirt_err"]) < 1e-5
        assert float(w["st_align_x"]) < 1e-12 and int(w["st_align_nfev"]) == 0
    assert int(eight["e_uploaded_rows"]) == 2 + 16 and int(one["e_uploaded_rows"]) == 10 + 16
    assert abs(eight["e_rmse"][0] / one["e_rmse"][0] - 1) < 1e-5 and np.max(np.abs(eight["e_xyz"] - one["e_xyz"])) < 0.2
    assert eight["e_shift_err"][-1] < 0.7 * np.abs(eight["e_true"][:, :2]).mean()
    # more ranks than angles (6 on 8): two ranks own nothing, issue every collective, and everybody ends with the one-rank answer
    assert int(eight["few_empty_ranks"]) == 2 and int(one["few_empty_ranks"]) == 0 and bool(eight["few_pipelined"])
    assert rel_max(eight["few_sirt_rec"], one["few_sirt_rec"]) < 1e-5 and np.allclose(eight["few_sirt_err"], one["few_sirt_err"], rtol=1e-5)
    assert rel_max(eight["few_cgls_rec"], one["few_cgls_rec"]) < 1e-5 and np.allclose(eight["few_cgls_err"], one["few_cgls_err"], rtol=1e-5)
    assert np.allclose(eight["few_align_x"], one["few_align_x"], atol=1e-12) and np.array_equal(eight["few_align_nfev"], one["few_align_nfev"])
    assert np.abs(one["few_align_x"]).max() > 0.1


def test_angle_split_is_the_reference_split():
    from tomography_alignment_amd.recon import sirt_mpi
    from tomography_alignment_amd.utilities.geometry import Geometry
    geo = Geometry(10, np.array([8, 8, 8]), np.ones(3), np.array([8, 8]), np.ones(2), cor_shift=np.arange(30.).reshape(10, 3))
    rows = [np.array_split(np.arange(10), 4)[r] for r in range(4)]           # recon/sirt_mpi.py:40
    assert [len(r) for r in rows] == [3, 3, 2, 2]
    sh = sirt_mpi.SIRT._shard_geometry(geo, rows[2])
    assert sh.n_proj == 2 and np.array_equal(sh.cor_shift, geo.cor_shift[6:8]) and geo.n_proj == 10


def test_id_rendezvous_between_processes():
    """The ncclUniqueId hand-off of RcclComm.from_env (rank 0 serves it on a loopback port, the others fetch it), with plain
    bytes -- and a listener left behind by an EARLIER launch (another key) on the first candidate port, which must be told
    apart and skipped (a stale id must never be read)."""
    from tomography_alignment_amd import comm as tcomm
    base = _free_port()
    stale = socket.socket()
    try:
        stale.bind(("127.0.0.1", tcomm.candidate_ports(base)[0]))
    except OSError:
        pytest.skip("candidate port taken")
    stale.listen(8)
    import threading

    def stale_server():                 # answers like a crashed launch's rank 0 would: with its own (old) id, for its own key only
        stale.settimeout(60)
        try:
            while True:
                c, _ = stale.accept()
                try:
                    n = int.from_bytes(c.recv(4), "little")
                    key = c.recv(n)
                    c.sendall(b"OK" + (128).to_bytes(4, "little") + bytes(128) if key == b"old" else b"NO")
                finally:
                    c.close()
        except OSError:
            pass
    th = threading.Thread(target=stale_server, daemon=True)
    th.start()
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "from tomography_alignment_amd.comm import exchange_from_rank0\n"
            "r = int(os.environ['RANK'])\n"
            "got = exchange_from_rank0(r, 3, lambda: bytes(range(128)), timeout=60, key='t', port=%d)\n"
            "assert got == bytes(range(128)); print('ok', r)\n") % (ROOT, base)
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(os.environ, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in (2, 1, 0)]      # the fetchers start first
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    stale.close()
    assert all(p.returncode == 0 for p in procs), outs
    assert sorted(o.strip() for o in outs) == ["ok 0", "ok 1", "ok 2"]
