#!/bin/bash
# usage (GPU box): tools/gpu_groups.sh <tag> -- GPU suite, then bench at C2 with pipelined slices (graph_loop line), alternating
tag=$1
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 4 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
run() { # name, env, args
  env $2 timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-variant $3 > gpurun_out/${tag}_$1.json 2> gpurun_out/${tag}_$1.err || { tail -n 20 gpurun_out/${tag}_$1.err; return 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$1.json"))
print("%-14s eager %.4f ms  graph %.4f ms  edge %.4f ms x%d" % ("$1", d["ms_per_step"], d["graph_loop"]["ms_per_step"], d["roofline"]["avg_launch_ms"], d["config"]["slices_per_gpu"]))
PY
}
for rep in 1 2; do
  run g1_$rep "A=1" "--groups 1" || exit 1
  run g2_$rep "A=1" "--groups 2" || exit 1
  run g2c256_$rep "ARREAU_GROUP_WGS=256" "--groups 2" || exit 1
  run g3_$rep "A=1" "--groups 3" || exit 1
  run g4_$rep "A=1" "--groups 4" || exit 1
  run g4c128_$rep "ARREAU_GROUP_WGS=128" "--groups 4" || exit 1
done
