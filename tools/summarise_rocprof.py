#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel-trace --stats CSV, --pmc CSVs) into the small tracked files under
profiles/.  Usage:
  tools/summarise_rocprof.py TAG STATS_DIR [FETCH_DIR WRITE_DIR] --workload "N=1024 n_proj=1024 n_gpus=1"
HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3): counters are in KB; on gfx950 FETCH_SIZE counts
exactly 1/2 of coalesced reads (re-calibrated here on kernels with a known byte count: k_pad, k_absmax,
k_dot read 4 GiB and report 2.097e6 KB), WRITE_SIZE is exact:  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024."""
import argparse
import csv
import glob
import json
import os


def _src_hash():
    """Hash of the kernel sources the counters were taken on (tomography_alignment_amd._lib.kernel_source_hash): bench.py refuses
    counters whose hash differs from the sources it runs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tomography_alignment_amd import _lib
    return _lib.kernel_source_hash()


ALIAS = {"k_adj_gather_flat<3>": "k_adj_gather_flat", "k_adj_gather_flat<6>": "k_adj_gather_flat", "k_tile_flat<true>": "k_fwd_tile_flat", "k_fwd_flat_z<2>": "k_fwd_tile_flat", "k_fwd_flat_z<1>": "k_fwd_tile_flat", "k_fwd_flat_z<2, 16>": "k_fwd_tile_flat", "k_fwd_flat_z<1, 16>": "k_fwd_tile_flat",
         "k_fwd_flat_z<1, 32>": "k_fwd_tile_flat", "k_fwd_flat_tab": "k_fwd_tile_flat", "k_tile_flat<false>": "k_adj_tile_flat", "k_tile<true>": "k_fwd_tile",
         "k_tile<false>": "k_adj_tile", "k_proj_grad<true>": "k_cost_grad(v1)", "k_proj_grad<false>": "k_proj_grad(v1)",
         "k_proj_grad_v2<true>": "k_cost_grad(v2)", "k_proj_grad_v2<false>": "k_proj_grad(v2)",
         "k_proj_grad_v3<true>": "k_cost_grad(v3)", "k_proj_grad_v3<false>": "k_proj_grad(v3)"}


def short(name):
    k = name.split("(")[0].replace("void ", "")
    if k.startswith("k_tile<true"):          # k_tile<FWD, waves per work-group> (round 5)
        return "k_fwd_tile"
    if k.startswith("k_tile<false"):
        return "k_adj_tile"
    return ALIAS.get(k, k)


def find(d, pat):
    m = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return m[0] if m else None


def merge_counters(path, tag, key, workload, kernels):
    """Counter files hold several workloads taken on ONE set of kernel sources: {"source", "src_hash", "workloads": {key: {"workload",
    "kernels"}}}.  A file of other sources (or of the round-3 single-workload layout) is replaced, one of the same sources is added to."""
    h = _src_hash()
    j = {}
    try:
        j = json.load(open(path))
    except (OSError, ValueError):
        pass
    if j.get("src_hash") != h or "workloads" not in j:
        j = {"source": tag, "src_hash": h, "workloads": {}}
    j["source"] = tag.split("_")[0] if j["workloads"] else tag
    j["workloads"][key] = {"workload": workload, "source": tag, "kernels": kernels}
    json.dump(j, open(path, "w"), indent=1)


def pmc(d, name):
    """kernel -> per-dispatch totals of counter `name`, in dispatch order"""
    out = {}
    if not d:
        return out
    path = find(d, "*counter_collection.csv")
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = (short(r["Kernel_Name"]), int(r["Dispatch_Id"]))
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    for (k, i) in sorted(per, key=lambda t: t[1]):
        out.setdefault(k, []).append(per[(k, i)])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("stats_dir")
    ap.add_argument("fetch_dir", nargs="?")
    ap.add_argument("write_dir", nargs="?")
    ap.add_argument("--workload", default="")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--key", default="", help="workload key bench.py matches, e.g. N1024_A1024_G1")
    ap.add_argument("--which", default="max", choices=["max", "last"],
                    help="which dispatch of a kernel the PMC table reports: the largest, or the last (= the timed step of a --steps 1 --warmup 1 run)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    stats_csv = find(a.stats_dir, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(stats_csv))) if stats_csv else []       # (a PMC-only call of a round has no stats pass of its own)
    fetch, write = pmc(a.fetch_dir, "FETCH_SIZE"), pmc(a.write_dir, "WRITE_SIZE")
    lines = ["# rocprofv3 summary `%s`" % a.tag, "", "workload: %s" % a.workload, "",
             "## `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`", "",
             "| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %s | %.3f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6,
                                                          float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    # per-dispatch durations: the stats above average over EVERY dispatch of the process (set-up launches on other inputs
    # included); bench.py's `kernels` / `roofline.avg_launch_ms` average the timed steps only = the last dispatches below
    trace = find(a.stats_dir, "*kernel_trace.csv")
    if trace:
        per = {}
        for r in csv.DictReader(open(trace)):
            per.setdefault(short(r["Kernel_Name"]), []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
        lines += ["", "## per-dispatch durations (ms, in launch order) of the projector kernels", ""]
        for k, v in per.items():
            if k.startswith(("k_fwd", "k_adj", "k_cost_grad")):
                lines.append("* `%s`: %s" % (k, ", ".join("%.1f" % x for x in v)))
    traffic = {}
    if fetch or write:
        pick = (lambda v: v[-1]) if a.which == "last" else max
        lines += ["", "## PMC passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs; per launch, %s dispatch of each kernel)" % a.which, "",
                  "| kernel | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes/launch = (2*F + W)*1024 |", "|---|---|---|---|"]
        for k in sorted(set(fetch) | set(write)):
            f = pick(fetch.get(k, [0.0]))
            w = pick(write.get(k, [0.0]))
            b = (2.0 * f + w) * 1024.0
            if b < 1e6:
                continue
            traffic[k] = {"fetch_kb": f, "write_kb": w, "hbm_bytes_per_launch": b}
            lines.append("| `%s` | %.4g | %.4g | %.4g |" % (k, f, w, b))
    open(os.path.join(a.out, a.tag + "_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
    if traffic:
        merge_counters(os.path.join(a.out, "pmc_traffic.json"), a.tag, a.key, a.workload, traffic)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
