export BENCH_ARGS="--workload dragon-sponza-1080p"
for v in ploc host karras; do
  case $v in ploc) unset RGK_BVH_BUILD RGK_LBVH_PLOC;; host) export RGK_BVH_BUILD=host;; karras) unset RGK_BVH_BUILD; export RGK_LBVH_PLOC=0;; esac
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 $BENCH_ARGS > gpurun_out/dragon_$v.json 2> gpurun_out/dragon_$v.err || { echo "failed $v"; tail -3 gpurun_out/dragon_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/dragon_$v.json"))
r=d["roofline"]
print("$v", d["value"], "ms/step", d["ms_per_step"], {k["kernel"]: k["ms_per_step"] for k in r["kernels"]})
PY
done
