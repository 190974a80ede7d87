#!/bin/bash
# round 4: boustrophedon traversal of the stash x load policy of the basis copies (nt / default), 256 x 20, alternating
tag=${1:-r04e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in cur nont; do
for b in 0 1; do
  if [ $lib = cur ]; then unset ARREAU_HIP_LIB; else export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/lib_$lib.so; fi
  ARREAU_CONV_PROJ_BOUSTROPHEDON=$b timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-fp32-variant --no-full-sampler --steps 60 > gpurun_out/${tag}_${lib}_b${b}_$i.json 2> gpurun_out/${tag}_${lib}_b${b}_$i.err || { tail -n 20 gpurun_out/${tag}_${lib}_b${b}_$i.err; exit 1; }
  python3 -c "
import json
d = json.load(open('gpurun_out/${tag}_${lib}_b${b}_$i.json')); r = d['roofline']
print('$lib', 'boustrophedon=$b', $i, 'ms_per_step', round(d['ms_per_step'], 4), 'eager', round(d['eager_loop']['ms_per_step'], 4), 'conv_proj us', round(1e3 * r['avg_launch_ms'], 1))"
done; done; done
