#!/bin/bash
# usage (GPU box): tools/gpu_configs.sh <tag>  -- GPU suite, then bench.py at every BASELINE configuration (c2 c1 c5 c3 c4)
tag=$1
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 4 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
for c in c2 c1 c5 c3 c4; do
  steps=30; [ $c = c1 ] && steps=99; [ $c = c4 ] && steps=10; [ $c = c5 ] && steps=20
  timeout -k 10 500 python bench.py --config $c --steps $steps > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || { tail -n 30 gpurun_out/${tag}_bench_$c.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_$c.json"))
r = d["roofline"]
print("$c", "ms_per_step", round(d["ms_per_step"], 4), "value", round(d["value"], 1), "graph", (d.get("graph_loop") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "frac", round(r["frac"], 4), "edge_ms", r.get("avg_launch_ms"), "fb_ms", d.get("forward_backward_ms"), "E/N", d["config"].get("edges_per_atom_end"))
PY
done
