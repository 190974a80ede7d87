#!/bin/bash
# usage (GPU box): tools/exp/c2_env_sweep.sh -- same-box pairs of the library's tunables at 256 x 20 (graph loop, 60 steps each)
cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-full-sampler --no-fp32-variant 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-44s %.4f ms  launch %.1f us' % ('$*', d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))"; }
for rep in 1 2; do
run X=default
run ARREAU_MLP_SLOTS=4
run ARREAU_CONV_PROJ_BOUSTROPHEDON=0
run ARREAU_EDGE_WGS=512
run ARREAU_MLP_NB=1
done
