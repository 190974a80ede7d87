#!/bin/bash
# usage (GPU box): tools/gpu_tests_from.sh <tag> "<pytest -k expression>" [prev-lib tag]  -- selected GPU tests, then (optionally) the A/B of
# tools/gpu_round.sh
tag=$1; kexpr=$2; prev=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -s -k "$kexpr" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
grep -n "passed\|failed" gpurun_out/${tag}_pytest.log | tail -n 2
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/${tag}_pytest.log | tail -n 20; exit $rc; }
[ -n "$prev" ] || exit 0
for i in 1 2; do for v in $prev cur; do
  if [ $v = cur ]; then unset ARREAU_HIP_LIB; else export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-fp32-variant --no-full-sampler --no-other-configs --steps 60 > gpurun_out/${tag}_ab_${v}_$i.json 2>gpurun_out/${tag}_ab_${v}_$i.err || { tail -n 20 gpurun_out/${tag}_ab_${v}_$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/${tag}_ab_${v}_$i.json')); print('$v', $i, 'ms_per_step', round(d['ms_per_step'],4), 'eager', round(d['eager_loop']['ms_per_step'],4), 'conv_proj us', round(1e3*d['roofline']['avg_launch_ms'],1), 'edge us', round(1e3*d['roofline']['edge_kernel'].get('avg_launch_ms', 0),1) if 'edge_kernel' in d['roofline'] else '')"
done; done
