#!/bin/bash
# usage (GPU box): tools/exp/ab_fuse_tail.sh -- the per-crystal tail launch (tail.hip) against the four launches it replaces, by batch size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "--config c1 --steps 99 --warmup 20" "--batch-per-gpu 4 --atoms 20 --steps 200 --warmup 20" "--batch-per-gpu 16 --atoms 20 --steps 200 --warmup 20" "--batch-per-gpu 64 --atoms 20 --steps 100 --warmup 10" "--batch-per-gpu 256 --atoms 20 --steps 60"; do
  for rep in 1 2; do for f in ${FUSE_VALUES:-0 1}; do
    ARREAU_FUSE_TAIL=$f timeout -k 10 300 python3 bench.py $cfg --no-cpu-baseline --no-fp32-variant --no-full-sampler --no-other-configs > gpurun_out/ft.json 2> gpurun_out/ft.err || { tail -n 20 gpurun_out/ft.err; exit 1; }
    python3 -c "import json; d=json.load(open('gpurun_out/ft.json')); print('$cfg | fuse_tail=$f rep $rep: eager', round(d['eager_loop']['ms_per_step'],4), 'graph', round(d['graph_loop']['ms_per_step'],4))"
  done; done
done
