#!/bin/bash
# On the GPU box: tools/micro/fetch_calib.bin (gathers with a known number of touched 128-byte lines) under the counters the
# HBM accounting uses -- FETCH_SIZE on its own, then the raw L2 memory-side request counters.  Prints, per kernel launch,
# requests per touched line and bytes per request under each reading.  Output: gpurun_out/fetch_calib.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/fcal_*
$R/tools/micro/fetch_calib.bin > $R/gpurun_out/fcal_plain.txt 2>&1 || { cat $R/gpurun_out/fcal_plain.txt; exit 1; }
i=0
for ctrs in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_MISS_sum" "TCC_REQ_sum TCC_HIT_sum TCC_READ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs -d $R/gpurun_out/fcal_pmc$i --output-format csv -- $R/tools/micro/fetch_calib.bin > $R/gpurun_out/fcal_pmc$i.txt 2>&1 || { tail -5 $R/gpurun_out/fcal_pmc$i.txt; exit 1; }
done
python3 - <<PY | tee $R/gpurun_out/fetch_calib.txt
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob("$R/gpurun_out/fcal_pmc*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
        rows.setdefault(k, {})[r["Counter_Name"]] = rows.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(open("$R/gpurun_out/fcal_plain.txt").read())
N = 16 * 2 ** 20
for (d, k), c in rows.items():
    if "stream" in k:
        lines = 2 ** 30 / 128
    elif k.startswith("pair"):
        lines = 131072
    elif k.startswith("evict"):
        lines = 8 * 200 * 64   # per first-touched line: 17 requests if the second touch hits in L2, 18 if not
    else:
        lines = N
    rd, r32, bub = c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_BUBBLE_sum", 0)
    print(f"{d:3d} {k:22s} lines {lines:.4g}  FETCH_SIZE {c.get('FETCH_SIZE', 0) * 1024:.4g} B = {c.get('FETCH_SIZE', 0) * 1024 / lines:6.1f} B/line   "
          f"RDREQ {rd:.4g} = {rd / lines:5.2f}/line  32B {r32:.4g}  BUBBLE {bub:.4g}  TCC_MISS {c.get('TCC_MISS_sum', 0):.4g}  TCC_REQ {c.get('TCC_REQ_sum', 0):.4g}  TCC_HIT {c.get('TCC_HIT_sum', 0):.4g}  TCP->TCC reads {c.get('TCP_TCC_READ_REQ_sum', 0):.4g}")
PY
