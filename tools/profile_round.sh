#!/bin/bash
# On the GPU box: the profiles a round commits under profiles/ -- kernel-trace stats of the default bench run, the two
# HBM-traffic counter passes and the VALU instruction-class passes (tools/pmc_roofline.txt); counters always on their own.
# usage: tools/profile_round.sh <tag> [bench args]; then locally: python tools/summarize_prof.py <tag>
# (<tag> = <round>_<workload>, e.g. r03_sponza-1080p: bench.py attaches profiles/*_<workload>_roofline.json to the same workload's line)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${tag}_*
echo "stats pass"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_stats --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_${tag}_stats.err || exit 1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "pmc pass $i: $line"
  timeout -k 10 400 rocprofv3 --pmc $line -d $R/gpurun_out/prof_${tag}_pmc$i --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/prof_${tag}_pmc$i.err || exit 1
done < <(echo FETCH_SIZE; echo WRITE_SIZE; echo "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; cat $R/tools/pmc_roofline.txt)
# the traces themselves are large: keep the stats tables, and of the counter tables (one row per dispatch and counter: tens of MB
# for a round of a thousand launches) the per-kernel sums -- gpurun brings back 64 MiB at most
find $R/gpurun_out/prof_${tag}_stats -name "*kernel_trace.csv" -delete
find $R/gpurun_out -path "*prof_${tag}_*" -name "*agent_info.csv" -delete
python3 - <<PY
import collections, csv, glob, os
for f in glob.glob("$R/gpurun_out/prof_${tag}_pmc*/**/*_counter_collection.csv", recursive=True):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        a = agg.setdefault((r["Kernel_Name"], r["Counter_Name"]), [0.0, set()])
        a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    with open(f.replace("_counter_collection.csv", "_counter_sums.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatches"])
        for (k, c), (v, d) in agg.items():
            w.writerow([k, c, repr(v), len(d)])
    os.remove(f)
PY
cut -c1-400 $R/gpurun_out/prof_${tag}_bench.json
