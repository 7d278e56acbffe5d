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
done <<'PASSES'
TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
PASSES
