for q in 4 5 6 8 10; do
  RGK_DEBUG_BVH=1 RGK_WALK_Q=$q timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/walk_$q.json 2> gpurun_out/walk_$q.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/walk_$q.json"))
r=d["roofline"]
print("walk_q=$q", d["value"], "trace_ms", r["avg_launch_ms"], "nodes", r["nodes_per_ray"], r["other_kernels_ms"])
PY
done
grep "rgk\]" gpurun_out/walk_0.err | head -2
