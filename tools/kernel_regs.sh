#!/bin/bash
# Registers / scratch / occupancy of every kernel, from the gfx950 assembly (no GPU needed): tools/kernel_regs.sh [extra hipcc flags]
out=${TMPDIR:-/tmp}/rgk_kernels.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -x hip --cuda-device-only -S "$@" \
    -I$(dirname $0)/../include $(dirname $0)/../rgk_amd/csrc/rgk_kernels.hip -o $out 2>/dev/null || exit 1
awk '/^_Z[0-9]+k_[a-z_]+/{n=$1; sub(/^_Z[0-9]+/,"",n); sub(/(8DevScene|10PassParams|ILb|Pj|j).*/,"",n); name=n}
     /; NumVgprs:/{v=$3} /; ScratchSize:/{s=$3} /; Occupancy:/{printf "%-24s vgprs %3d scratch %4d occupancy %d\n", name, v, s, $3}' $out
