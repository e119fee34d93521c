#!/usr/bin/env python3
"""Tabulate rocprofv3 --pmc passes: for each kernel (name substring) the per-dispatch MAX of every counter found under the
given directories.  usage: tools/pmc_table.py <kernel substring> dir1 [dir2 ...]"""
import csv, glob, os, sys
from collections import defaultdict
key = sys.argv[1]
vals = defaultdict(lambda: defaultdict(float))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if key in row["Kernel_Name"]:
                vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for name in sorted(vals):
    v = vals[name]
    print("%-44s max %.6g  (n_dispatch %d)" % (name, max(v.values()), len(v)))
