#!/bin/bash
# usage: tools/ab_env.sh "VAR=value ..." [bench args]  (GPU box): bench with and without the given environment, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
envs="$1"; shift
for rep in 1 2; do
  for which in base alt; do
    if [ $which = alt ]; then pre="env $envs"; else pre=""; fi
    $pre timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/abe_${which}_$rep.json 2> gpurun_out/abe_${which}_$rep.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/abe_${which}_$rep.json"))
print("%-5s rep $rep: step %.3f ms  edge %.3f ms" % ("$which", d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
