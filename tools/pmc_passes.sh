#!/bin/bash
# usage: tools/pmc_passes.sh <tag> [bench args...]   -- separate rocprofv3 --pmc passes (counters only) of one bench round
# run on the GPU box; writes gpurun_out/pmc_<tag>_<pass>/ and prints the per-kernel averages
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  out=$R/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  echo "pass $i: $line"
  timeout -k 10 240 rocprofv3 --pmc $line -d $out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $out.json 2> $out.err
  python3 $R/tools/pmc_summary.py $out | grep -v "k_init\|k_build\|k_resolve\|k_raygen" 
done < <(cat ${PASSES:-$R/tools/pmc_default.txt})
