"""bench.py prints ONE JSON line with the driver's contract fields plus a compact `roofline` and `cpu_baseline` -- under 4 KB, whatever legs
ran -- and writes everything it measured to the `--detail` file (checked on a tiny workload so the test takes seconds; the line's size at
the full-size counters is held by tests/test_host_logic.py::test_bench_contract_line_stays_small_at_the_full_size_counters)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra, want_line=False):
    """(the full record of the --detail file) or (stdout line, record)."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "detail.json")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "64", "--angles", "48", "--steps", "2", "--warmup", "1", "--detail", path] + extra,
                             capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and len(lines[0]) < 4096, out.stdout
        line, detail = json.loads(lines[0]), json.load(open(path))
    assert line["detail"] == path and line["value"] == detail["value"] and line["ms_per_step"] == detail["ms_per_step"]
    return (line, detail) if want_line else detail


def test_bench_json_contract():
    line, d = _run([], want_line=True)
    # the stdout line: the driver's keys, typed, and the compact blocks
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None and set(line["config"]) == {"workload", "sharding"}
    lr = line["roofline"]
    assert lr["bound"] == "hbm" and lr["peak"] == 8000.0 and abs(lr["frac"] - lr["achieved"] / lr["peak"]) < 1e-3 and lr["counters"] is None and lr["avg_launch_ms"] > 0
    lc = line["cpu_baseline"]
    assert lc["kind"] == "port" and lc["cores"] == 1 and lc["value"] > 0 and isinstance(lc["sample"], str)
    assert line["value_dense_volume"] > 0 and line["value_tilted_poses"] > 0 and line["cgls_it_per_s"] > 0 and line["alignment_gradient"]["evals_per_sec"] > 0
    assert line["align_rigid_e2e"]["wall_s"] > 0 and sum(line["kernel_ms_per_step"].values()) <= line["ms_per_step"]
    # the record behind it
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "it/s" and d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-2
    r = d["roofline"]
    # no committed PMC counters exist for this tiny workload: the roofline falls back to the algorithmic-HBM figure and says so;
    # at the headline size `bound` is the busiest of hbm / valu_issue / lds from profiles/ (tests/test_host_logic.py checks that path)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel"].startswith(("k_fwd", "k_adj")) and r["achieved"] > 0 and "traffic" in r and r["counters"] is None and "hbm_algorithmic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "it/s" and c["value"] > 0 and isinstance(c["sample"], str)
    assert d["value"] > 100 * c["value"]
    a = d["alignment_gradient"]
    assert a["evals_per_sec"] > 0 and a["dense_volume"]["evals_per_sec"] > 0
    assert d["dense_volume"]["value"] > 0 and d["tilted_poses"]["value"] > 0
    # config 5 end to end: two outer iterations, the second one's SIRT at the recovered (tilted) poses
    e = d["align_rigid_e2e"]
    assert len(e["outer"]) == 2 and e["sharded_code_path"] is False and e["outer"][1]["sirt_wall_s"] > 0 and e["alignment_evals"] > 0
    assert d["cgls"]["value"] > 0 and d["cgls"]["pipelined"] is False


def test_bench_gpus_2_on_a_one_gpu_box_fails_cleanly():
    """`python bench.py --gpus N` launches its own ranks; with fewer GPUs than ranks it must say so and
    exit non-zero before starting anything."""
    import ctypes
    from tomography_alignment_amd import _lib
    n = ctypes.c_int(0)
    _lib.load().tomo_device_count(ctypes.byref(n))
    if n.value >= 2:
        pytest.skip("%d GPUs visible" % n.value)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--angles", "48"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 2 and "2 GPUs requested, %d visible" % n.value in out.stderr and out.stdout.strip() == ""


def test_bench_sharded_legs_print_the_same_keys_as_the_plain_run():
    """`bench.py --gpus 1 --force-sharded` runs every leg a `--gpus 8` run will -- SIRT, dense / tilted side runs, CGLS
    (sharded, pipelined), alignment gradient, align_rigid end to end through sirt_mpi.SIRT + align_projections_sharded -- and prints the
    key set of the plain run."""
    plain = _run(["--no-cpu-baseline"])
    shard = _run(["--no-cpu-baseline", "--force-sharded"])
    assert set(shard) == set(plain), set(shard) ^ set(plain)
    e = shard["align_rigid_e2e"]
    assert e["sharded_code_path"] is True and e["ranks"] == 1 and len(e["outer"]) == 2 and set(e) == set(plain["align_rigid_e2e"])
    assert shard["cgls"]["pipelined"] is True and shard["cgls"]["value"] > 0 and "reduce_scatter_f32_ms_per_step" in shard["cgls"]
    assert "reduce_scatter_f32" in shard["kernels"] and "reduce_scatter_f32" not in plain["kernels"]
    # the sharded loop at world 1 = the plain loop up to float32 sums in another order (first outer iteration: before the optimiser's
    # sensitivity enters, tests/test_dist_gloo.py)
    assert abs(e["outer"][0]["rmse_after_sirt"] / plain["align_rigid_e2e"]["outer"][0]["rmse_after_sirt"] - 1) < 1e-4


def test_bench_sharded_code_path_on_one_gpu():
    d = _run(["--force-sharded", "--no-cpu-baseline", "--no-align", "--perturbed"])
    assert d["value"] > 0 and d["kernels"]["k_adj_tile"]["launches_per_step"] >= 2      # x-slab pipelined back-projection
