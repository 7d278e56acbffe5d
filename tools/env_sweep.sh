# usage: tools/env_sweep.sh "VAR=a" "VAR=b OTHER=c" ...  -- bench the product library under each environment
for e in "$@"; do
  tag=$(echo "$e" | tr ' =/' '___')
  env RGK_DEBUG_BVH=1 $e timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/env_$tag.json 2> gpurun_out/env_$tag.err || { echo "$e failed"; tail -3 gpurun_out/env_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/env_$tag.json"))
r=d["roofline"]
print("$e", d["value"], "ms/step", d["ms_per_step"], "nodes", r["whole_round"]["nodes_per_path_ray"], "tris", r["whole_round"]["tris_per_path_ray"], {k["kernel"]: k["ms_per_step"] for k in r["kernels"]})
PY
  grep "rgk\] bvh4" gpurun_out/env_$tag.err | head -1
done
