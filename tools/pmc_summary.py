#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_summary.py <dir>..."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        vals = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            vals[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k in vals:
            if k.startswith("k_"):
                n = len(disp[k])
                print(f"{k:32s} launches {n:3d} " + " ".join(f"{c}={v / n:.4g}" for c, v in sorted(vals[k].items())))
