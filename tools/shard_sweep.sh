# usage: tools/shard_sweep.sh <variant|base> K [K ...]  -- bench one rank's share (1/K of the tiles) of the default workload
v=$1; shift
if [ "$v" = base ]; then unset RGK_LIB; else export RGK_LIB=$PWD/rgk_amd/csrc/librgk_var_$v.so; fi
for k in "$@"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 6 --emulate-shard $k > gpurun_out/shard_${v}_$k.json 2> gpurun_out/shard_${v}_$k.err || { echo "failed"; tail -3 gpurun_out/shard_${v}_$k.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/shard_${v}_$k.json")); r=d["roofline"]
print("variant $v shard 1/$k", d["ms_per_step"], "ms", {x["kernel"]: x["ms_per_step"] for x in r["kernels"]})
PY
done
