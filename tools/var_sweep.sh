# usage: tools/var_sweep.sh a b c ...   -- bench each tuning variant librgk_var_<x>.so ("base" = product library)
for v in "$@"; do
  if [ "$v" = base ]; then unset RGK_LIB; else export RGK_LIB=$PWD/rgk_amd/csrc/librgk_var_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 $BENCH_ARGS > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/var_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/var_$v.json"))
r=d["roofline"]
print("variant $v", d["value"], "ms/step", d["ms_per_step"], "nodes", r["whole_round"]["nodes_per_path_ray"], {k["kernel"]: k["ms_per_step"] for k in r["kernels"]})
PY
done
