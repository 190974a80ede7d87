#!/bin/bash
# usage (GPU box): tools/gpu_ab.sh <tag> "VAR=value ..."  -- GPU tests, then an A/B of the default build against the given environment
tag=$1; envs="$2"
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 6 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
tools/ab_env.sh "$envs" --no-fp32-variant || exit 1
timeout -k 10 300 python bench.py --config c1 --steps 200 --warmup 20 --no-cpu-baseline --no-fp32-variant > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err || { tail -n 30 gpurun_out/${tag}_bench_c1.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_c1.json"))
print("c1 ms_per_step", round(d["ms_per_step"], 4), "free_loop", d.get("free_running_loop"))
PY
