#!/bin/bash
# On the GPU box: kernel-trace stats of the default bench run + the two HBM-traffic counter passes (counters on their own).
# usage: tools/profile_round.sh <tag>; then locally: python tools/summarize_prof.py <tag> gpurun_out/prof_<tag>_stats gpurun_out/prof_<tag>_fetch gpurun_out/prof_<tag>_write
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for d in stats fetch write; do rm -rf $R/gpurun_out/prof_${tag}_$d; done
echo "stats pass"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_stats --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_${tag}_stats.err || exit 1
echo "fetch pass"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_${tag}_fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_${tag}_fetch.err || exit 1
echo "write pass"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_${tag}_write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_${tag}_write.err || exit 1
# the kernel trace itself is large: keep only the stats tables
find $R/gpurun_out/prof_${tag}_stats -name "*kernel_trace.csv" -delete
cat $R/gpurun_out/prof_${tag}_bench.json | cut -c1-300
