#!/bin/bash
# usage (GPU box): tools/exp/env_configs.sh "<configs>" "VAR=value" -- ms per step with and without an environment setting, alternating
cd $GRAFT_REPO_ROOT
for cfg in $1; do
  for which in base env base env; do
    if [ $which = env ]; then export $2; else unset ${2%%=*}; fi
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-variant > gpurun_out/envc_${cfg}_$which.json 2> gpurun_out/envc_${cfg}_$which.err || { tail -n 5 gpurun_out/envc_${cfg}_$which.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/envc_${cfg}_$which.json')); print('$cfg $which', round(d['ms_per_step'],4))"
  done
done
