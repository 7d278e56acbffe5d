#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the small files kept under profiles/.

  python tools/summarize_prof.py <round-tag> <stats_dir> [<fetch_dir> <write_dir>]
Writes profiles/<tag>_kernel_stats.csv (the --stats table, kernel names shortened) and
profiles/<tag>_hbm_traffic.csv (per kernel: launches, FETCH_SIZE / WRITE_SIZE sums in KB as
reported, and bytes per launch with the gfx950 correction FETCH x2 of MI355X_MICROARCH.md).
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def find(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return f[0] if f else None


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = find(stats_dir, "_kernel_stats.csv")
    with open(ks) as f, open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as g:
        w = csv.writer(g)
        for i, row in enumerate(csv.reader(f)):
            if i:
                row[0] = short(row[0])
            w.writerow(row)
    if len(sys.argv) >= 5:
        agg = collections.OrderedDict()
        for d, ctr in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
                if r["Counter_Name"] != ctr:
                    continue
                k = short(r["Kernel_Name"])
                a = agg.setdefault(k, {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
                a[ctr][0] += 1
                a[ctr][1] += float(r["Counter_Value"])
        with open(os.path.join(out, f"{tag}_hbm_traffic.csv"), "w", newline="") as g:
            w = csv.writer(g)
            w.writerow(["kernel", "launches", "FETCH_SIZE_KB_sum", "WRITE_SIZE_KB_sum", "read_MB_per_launch_x2corrected",
                        "write_MB_per_launch", "hbm_MB_per_launch"])
            for k, a in agg.items():
                n = max(a["FETCH_SIZE"][0], a["WRITE_SIZE"][0], 1)
                rd = 2.0 * a["FETCH_SIZE"][1] * 1024 / n / 1e6
                wr = a["WRITE_SIZE"][1] * 1024 / max(a["WRITE_SIZE"][0], 1) / 1e6
                w.writerow([k, n, round(a["FETCH_SIZE"][1], 1), round(a["WRITE_SIZE"][1], 1), round(rd, 2), round(wr, 2),
                            round(rd + wr, 2)])


if __name__ == "__main__":
    main()
