#!/usr/bin/env python3
"""Condense rocprofv3 --pmc SQ passes of a bench.py run into profiles/sq_counters.json (read back by bench.py's roofline) and
a markdown table.  For every projector kernel the LAST dispatch of the run is taken (= the timed SIRT step of
`bench.py --steps 1 --warmup 1`: earlier dispatches are set-up launches on other inputs), counters summed over that dispatch.

  tools/summarise_sq.py TAG KEY DIR [DIR ...] [--out profiles]

Derived: lds_bytes = SQ_INSTS_LDS x the bytes one LDS wave-instruction of that kernel moves (table below, from the kernel
source: the flat forward's LDS traffic is ds_read2st64_b32 = 2 dwords x 64 lanes; the gather adjoint's ds_read_b32 /
ds_write_b32 = 1 dword x 64 lanes; the general tile kernels' ds_read2_b32 = 2 dwords, ds_add_u32 = 1 dword)."""
import argparse
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarise_rocprof import short, _src_hash, merge_counters  # noqa: E402,F401

LDS_BYTES_PER_INSTR = {"k_fwd_tile_flat": 512, "k_adj_gather_flat": 256, "k_fwd_tile": 512, "k_adj_tile": 256, "k_adj_tile_flat": 256}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("key")
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--workload", default="")
    a = ap.parse_args()
    per = {}                                        # kernel -> dispatch id -> counter -> value
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k.startswith(("k_fwd", "k_adj", "k_cost_grad")):
                    continue
                per.setdefault(k, {}).setdefault((os.path.dirname(f), int(r["Dispatch_Id"])), {})
                c = per[k][(os.path.dirname(f), int(r["Dispatch_Id"]))]
                c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    kernels = {}
    lines = ["# SQ counters `%s`" % a.tag, "", "workload: %s (key `%s`); last dispatch of each kernel = the timed step" % (a.workload, a.key), ""]
    for k, disp in sorted(per.items()):
        merged = {}
        for d in set(dd for dd, _ in disp):         # one pass directory at a time: its last dispatch of this kernel
            last = max(i for dd, i in disp if dd == d)
            merged.update(disp[(d, last)])
        if "SQ_INSTS_LDS" in merged and k in LDS_BYTES_PER_INSTR:
            merged["lds_bytes"] = merged["SQ_INSTS_LDS"] * LDS_BYTES_PER_INSTR[k]
            merged["lds_bytes_per_instr_assumed"] = LDS_BYTES_PER_INSTR[k]
        kernels[k] = merged
        lines += ["## `%s`" % k, "", "| counter | per launch |", "|---|---|"] + ["| `%s` | %.6g |" % (n, v) for n, v in sorted(merged.items())] + [""]
    os.makedirs(a.out, exist_ok=True)
    merge_counters(os.path.join(a.out, "sq_counters.json"), a.tag, a.key, a.workload, kernels)
    open(os.path.join(a.out, a.tag + "_sq_counters.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
