#!/bin/bash
# usage (GPU box): tools/exp/ab_configs.sh "<configs>" -- ms per step of the previous build (tools/exp/ab/lib_prev.so) and the in-tree one
cd $GRAFT_REPO_ROOT
for cfg in $1; do
  for which in prev cur prev cur; do
    if [ $which = prev ]; then export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/${AB_LIB:-lib_prev.so}; else unset ARREAU_HIP_LIB; fi
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-variant > gpurun_out/abc_${cfg}_$which.json 2> gpurun_out/abc_${cfg}_$which.err || { tail -n 5 gpurun_out/abc_${cfg}_$which.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/abc_${cfg}_$which.json')); print('$cfg $which', round(d['ms_per_step'],4), 'edge', round(d['roofline']['avg_launch_ms'],4) if 'roofline' in d and d['roofline'].get('avg_launch_ms') else '')"
  done
done
