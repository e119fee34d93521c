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


def _run(world, out):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return np.load(out)


@pytest.mark.timeout(900)
def test_sharded_solvers_match_unsharded(tmp_path):
    one = _run(1, str(tmp_path / "w1.npz"))
    two = _run(2, str(tmp_path / "w2.npz"))
    assert rel_max(two["rec"], one["rec"]) < 1e-5           # float32 sums in a different order, nothing else
    assert np.allclose(two["err"], one["err"], rtol=1e-5)
    assert rel_max(two["crec"], one["crec"]) < 1e-4
    assert np.allclose(two["cerr"], one["cerr"], rtol=1e-4)
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
