#!/bin/bash
# usage (GPU box): tools/exp/ab_loop_prep.sh -- the sampling loop without a prep launch per step (default) against ARREAU_LOOP_PREP=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "--config c1 --steps 99" "--steps 60" "--config c3 --steps 30"; do
  for f in 0 1 0 1; do
    ARREAU_LOOP_PREP=$f python3 bench.py $cfg --no-cpu-baseline --no-full-sampler --no-fp32-variant > gpurun_out/lp.json 2> gpurun_out/lp.err || { tail -n 5 gpurun_out/lp.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/lp.json')); print('$cfg', 'prep_per_step=$f', 'eager', round(d['eager_loop']['ms_per_step'],4), 'graph', round((d.get('graph_loop') or {}).get('ms_per_step',0),4))"
  done
done
