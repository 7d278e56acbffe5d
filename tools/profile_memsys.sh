#!/bin/bash
# On the GPU box: memory-system and LDS counters of one bench round (tools/pmc_memsys.txt), one rocprofv3 --pmc pass per line.
# usage: tools/profile_memsys.sh <tag> [bench args]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/mem_${tag}_*
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "pmc pass $i: $line"
  timeout -k 10 400 rocprofv3 --pmc $line -d $R/gpurun_out/mem_${tag}_pmc$i --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/mem_${tag}_pmc$i.err || { tail -5 $R/gpurun_out/mem_${tag}_pmc$i.err; exit 1; }
done < $R/tools/pmc_memsys.txt
find $R/gpurun_out -path "*mem_${tag}_*" -name "*agent_info.csv" -delete
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/gpurun_out/mem_${tag}_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    if "shade" in k or "trace" in k or "resolve" in k:
        print(k)
        for c in sorted(agg[k]): print(f"   {c:40s} {agg[k][c] / max(1, cnt[k][c]):.4g} per launch x {cnt[k][c]} rows")
PY
