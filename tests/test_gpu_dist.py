"""World size 2 ON THE GPU: two processes share GPU 0, each with its own context and the real HIP backend; the volume all-reduces
go through the host over torch.distributed/gloo (tests/_gloo_gpu_worker.py explains why not RCCL).  The angle-sharded SIRT --
slab pipeline and plain sequence -- must give the world-size-1 answer."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_max

pytestmark = pytest.mark.gpu


def _run(world, out):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_gpu_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=300)[0].decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                      # exactly the processes started here
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return np.load(out)


@pytest.mark.timeout(900)
def test_sharded_sirt_world_2_on_the_gpu(tmp_path):
    one = _run(1, str(tmp_path / "g1.npz"))
    two = _run(2, str(tmp_path / "g2.npz"))
    for tag in ("flat", "tilted"):
        ref = one["%s_plain_rec" % tag]
        for w, name in ((one, "world 1"), (two, "world 2")):
            for mode in ("pipelined", "allreduce", "plain"):
                assert rel_max(w["%s_%s_rec" % (tag, mode)], ref) < 1e-5, (tag, name, mode)
                assert np.allclose(w["%s_%s_err" % (tag, mode)], one["%s_plain_err" % tag], rtol=1e-5), (tag, name, mode)
            assert bool(w["%s_pipelined_pipelined" % tag]) and bool(w["%s_allreduce_pipelined" % tag]) and not bool(w["%s_plain_pipelined" % tag])
        # 5 iterations (counted after the constructor's all-reduce of V): pipelined = per iteration 6 slabs, each a reduce-scatter (a rank
        # receives and updates its own half; the other half is left NaN by the stand-in communicator) and an all-gather, no all-reduce
        # of a slab (8000-voxel planes split evenly) or of the volume; "allreduce" = round 3's 6 slab all-reduces per iteration;
        # plain = one whole-volume all-reduce per iteration
        assert int(two["%s_pipelined_nvol" % tag]) == 0 and int(two["%s_pipelined_nslab" % tag]) == 0
        assert int(two["%s_pipelined_nrs" % tag]) == 5 * 6 and int(two["%s_pipelined_nag" % tag]) == 5 * 6
        assert int(two["%s_allreduce_nvol" % tag]) == 0 and int(two["%s_allreduce_nslab" % tag]) == 5 * 6 and int(two["%s_allreduce_nrs" % tag]) == 0
        assert int(two["%s_plain_nvol" % tag]) == 5 and int(two["%s_plain_nslab" % tag]) == 0
        assert one["%s_plain_err" % tag][-1] < one["%s_plain_err" % tag][0]
        # CGLS (round 5): pipelined (reduce-scatter + all-gather per slab: 6 slabs x 5 iterations each), all-reduce form, plain -- world 2 = world 1
        cref, cerr_ref = one["%s_cgls_plain_rec" % tag], one["%s_cgls_plain_err" % tag]
        for w, name in ((one, "world 1"), (two, "world 2")):
            for mode in ("pipelined", "allreduce", "plain"):
                assert rel_max(w["%s_cgls_%s_rec" % (tag, mode)], cref) < 1e-5, (tag, name, mode, rel_max(w["%s_cgls_%s_rec" % (tag, mode)], cref))
                assert np.allclose(w["%s_cgls_%s_err" % (tag, mode)], cerr_ref, rtol=1e-5), (tag, name, mode)
        assert list(two["%s_cgls_pipelined_counts" % tag]) == [30, 30, 0, 1] and list(two["%s_cgls_allreduce_counts" % tag]) == [0, 0, 30, 1]
        assert list(two["%s_cgls_plain_counts" % tag]) == [0, 0, 0, 0] and cerr_ref[-1] < cerr_ref[0]
    # examples/align_rigid.run(comm=...) on this world: each half of each outer iteration against the unsharded loop on
    # the same inputs -- sharded SIRT (slab pipeline forced at world 1 too) at 1e-5; the alignment pass of a rank's own projections against
    # the replicated reconstruction: EXACTLY the unsharded pass (round 6: the fused cost / gradient reduction adds its work-group partials in a
    # fixed order, so a pose's evaluations do not depend on the launch they are part of; until round 5 float64 atomics made two identical
    # passes differ by 1.3e-4 px and this bound was 5e-3)
    for w, name in ((one, "world 1"), (two, "world 2")):
        for stage in (0, 1):
            assert float(w["st_sirt%d_rec" % stage]) < 1e-5 and float(w["st_sirt%d_err" % stage]) < 1e-5, (name, stage, float(w["st_sirt%d_rec" % stage]))
            assert float(w["st_eval%d_cost" % stage]) == 0.0 and float(w["st_eval%d_grad" % stage]) == 0.0, (name, stage, float(w["st_eval%d_grad" % stage]))
            assert float(w["st_align%d_x" % stage]) == 0.0 and float(w["st_align%d_fun" % stage]) == 0.0, (name, stage, float(w["st_align%d_x" % stage]))
            assert float(w["st_repeat%d_x" % stage]) == 0.0, (name, stage)          # the unsharded pass against itself
            print("align_rigid halves, %s, outer %d: SIRT rec %.1e err %.1e; evaluations cost %.1e grad %.1e; pass outcome table %.1e fun %.1e"
                  % (name, stage, float(w["st_sirt%d_rec" % stage]), float(w["st_sirt%d_err" % stage]), float(w["st_eval%d_cost" % stage]),
                     float(w["st_eval%d_grad" % stage]), float(w["st_align%d_x" % stage]), float(w["st_align%d_fun" % stage])))
        assert bool(w["e_pipelined"]) and float(w["st_pose_moved"]) > 0.5 and float(w["e_spread"]) == 0.0
        assert abs(w["e_rmse"][0] / one["e_rmse"][0] - 1) < 1e-5
        assert w["e_shift_err"][-1] < w["e_shift_err"][0] < 0.7 * float(w["e_injected"])
    # one projection on two ranks: rank 1 owns no angle (zero-row tables on the real backend, no projector call, every collective issued)
    assert int(two["one_angle_empty_ranks"]) == 1 and int(one["one_angle_empty_ranks"]) == 0
    for mode in ("pipelined", "plain"):
        assert rel_max(two["one_angle_%s_rec" % mode], one["one_angle_plain_rec"]) < 1e-5, mode
        assert np.allclose(two["one_angle_%s_err" % mode], one["one_angle_plain_err"], rtol=1e-5), mode
    assert one["one_angle_plain_rec"].max() > 0
