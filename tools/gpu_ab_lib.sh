#!/bin/bash
# usage (GPU box): tools/gpu_ab_lib.sh <tag of tools/exp/ab/lib_<tag>.so> [pytest -k expression]  -- a kernel change against the build before it:
# the tests that pin the kernel, then <tag> / in-tree alternating at 256 x 20 (graph replay ms per step + the message kernel's launch time)
prev=$1; kexpr=${2:-"batch_independence or launch_geometry or loop_at_the_benchmark or range_launches or many_ragged or plain_tolerance"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$kexpr" > gpurun_out/ab_${prev}_pytest.log 2>&1; rc=$?
tail -n 2 gpurun_out/ab_${prev}_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/ab_${prev}_pytest.log | tail -n 20; exit $rc; }
for i in 1 2; do for v in $prev cur; do
  if [ $v = cur ]; then unset ARREAU_HIP_LIB; else export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-fp32-variant --no-full-sampler --steps 60 > gpurun_out/ab_${v}_$i.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/ab_${v}_$i.json')); print('$v', $i, 'ms_per_step', round(d['ms_per_step'],4), 'eager', round(d['eager_loop']['ms_per_step'],4), 'conv_proj us', round(1e3*d['roofline']['avg_launch_ms'],1))"
done; done
