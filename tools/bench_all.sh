#!/bin/bash
# On the GPU box: the bench line of every BASELINE configuration that fits one GPU -> gpurun_out/bench_<workload>.json (copied to profiles/<round>_bench_*.json)
timeout -k 10 500 python bench.py > gpurun_out/bench_sponza-1080p.json 2> gpurun_out/bench_sponza-1080p.err || { tail -5 gpurun_out/bench_sponza-1080p.err; exit 1; }
for w in cornell-1024 dragon-sponza-1080p sponza4-2160p; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || { tail -5 gpurun_out/bench_$w.err; exit 1; }
done
python - <<PY
import json
for w in ("sponza-1080p", "cornell-1024", "dragon-sponza-1080p", "sponza4-2160p"):
    d = json.load(open(f"gpurun_out/bench_{w}.json")); r = d["roofline"]
    print(w, d["value"], d["unit"], d["ms_per_step"], "ms; first round", r.get("first_round_ms"), "; step_hbm_frac", r.get("step_hbm_frac"), "; line:", r.get("kernel"), r.get("bound"), r.get("frac"), "; cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
