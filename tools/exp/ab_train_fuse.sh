#!/bin/bash
# usage (GPU box): tools/exp/ab_train_fuse.sh [steps] -- the training step (bench.py --config c5) with and without the round-5 launch
# merges (ARREAU_TRAIN_FUSE=0), alternating runs in one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
steps=${1:-60}
for rep in 1 2 3; do
  for which in fuse plain; do
    unset ARREAU_TRAIN_FUSE ARREAU_TRAIN_SIDE_LATE; if [ $which = plain ]; then export ARREAU_TRAIN_FUSE=0; fi; if [ $which = early ]; then export ARREAU_TRAIN_SIDE_LATE=0; fi
    timeout -k 10 300 python3 bench.py --config c5 --steps $steps --warmup 10 --no-cpu-baseline > gpurun_out/abtf_${which}_$rep.json 2> gpurun_out/abtf_${which}_$rep.err || { tail -n 5 gpurun_out/abtf_${which}_$rep.err; exit 1; }
    python3 - <<PY
import json
d=json.load(open("gpurun_out/abtf_${which}_$rep.json"))
print("%-5s rep $rep: step %.4f ms  forward+backward %.4f ms" % ("$which", d["ms_per_step"], d["forward_backward_ms"]))
PY
  done
done
