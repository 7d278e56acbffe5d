#!/bin/bash
# On the GPU box: the calibration stream of tools/micro/valu_peak.hip under the SQ counters the roofline uses, one kernel per
# instruction class -- what SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE read on a known stream.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_calib
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_calib --output-format csv -- $R/tools/micro/valu_peak.bin > $R/gpurun_out/pmc_calib.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_calib/**/*_counter_collection.csv", recursive=True)[0]
v = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    v[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in v.items():
    last = {n: x[-1] for n, x in c.items()}   # the timed (second) launch
    insts = last.get("SQ_INSTS_VALU", 0)
    if not insts: continue
    cyc = last["GRBM_GUI_ACTIVE"] / 8
    print(f"{k[:40]:40s} INSTS_VALU {insts:.4g}  ACTIVE_INST_VALU/INSTS {last['SQ_ACTIVE_INST_VALU'] / insts:.3f}  cycles/inst/SIMD {cyc * 1024 / insts:.2f}  "
          f"FMA_F32/INSTS {last.get('SQ_INSTS_VALU_FMA_F32', 0) / insts:.2f} INT32/INSTS {last.get('SQ_INSTS_VALU_INT32', 0) / insts:.2f}  THREAD_CYCLES/(64 INSTS) {last.get('SQ_THREAD_CYCLES_VALU', 0) / 64 / insts:.3f}")
PY
