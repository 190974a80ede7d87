#!/bin/bash
# usage (GPU box): tools/exp/ab_fuse_small.sh -- one launch per layer (default) against two (ARREAU_FUSE_SMALL=0) at small batches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_launch or ragged or own_neighbor or sample_loop or graph_replay" 2>&1 | tail -n 3 || exit 1
for cfg in "--config c1 --steps 99" "--batch-per-gpu 4 --atoms 20 --steps 100" "--batch-per-gpu 12 --atoms 20 --steps 100" "--batch-per-gpu 25 --atoms 20 --steps 100"; do
  for f in 1 0 1 0; do
    ARREAU_FUSE_SMALL=$f python3 bench.py $cfg --no-cpu-baseline --no-full-sampler --no-fp32-variant > gpurun_out/fs.json 2> gpurun_out/fs.err || { tail -n 5 gpurun_out/fs.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/fs.json')); print('$cfg', 'fuse=$f', 'eager', round(d['eager_loop']['ms_per_step'],4), 'graph', round((d.get('graph_loop') or {}).get('ms_per_step',0),4))"
  done
done
