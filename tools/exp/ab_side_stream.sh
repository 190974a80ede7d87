#!/bin/bash
# GPU box: the training step (bench.py --config c5) with the fiber branch in line (ARREAU_TRAIN_SIDE_STREAM=0) and on its side stream,
# alternating on one box.  (Round 5: 2.1285 2.1345 2.1264 in line; 2.0913 2.1091 2.1226 side stream; 2.1152 2.1128 2.1146 with a
# lowest-priority side stream, an option since removed.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in "0 0" "1 0"; do
  set -- $v
  ARREAU_TRAIN_SIDE_STREAM=$1 ARREAU_TRAIN_SIDE_PRIORITY=$2 timeout -k 10 300 python3 bench.py --config c5 --no-cpu-baseline --steps 60 > gpurun_out/abss_$1$2_$i.json 2> gpurun_out/abss_$1$2_$i.err || { tail -n 20 gpurun_out/abss_$1$2_$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/abss_$1$2_$i.json')); print('side_stream=$1 low_priority=$2', $i, 'ms_per_step', round(d['ms_per_step'],4), 'fwd+bwd', round(d['forward_backward_ms'],4))"
done; done
