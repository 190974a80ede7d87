#!/bin/bash
# round 4: ConvNext kernel change -- the tests that pin it bit for bit, then prev / new alternating (kernel averages from rocprofv3)
tag=${1:-r04k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_launch or launch_geometry or full_size_architecture or plain_tolerance or loop_at_the_benchmark or batch_independence or teacher_forced" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/${tag}_pytest.log | tail -n 30; exit $rc; }
tools/exp/sweep_libs.sh "prevmlp" --no-full-sampler
for i in 1 2; do for v in prevmlp cur; do
  if [ $v = cur ]; then unset ARREAU_HIP_LIB; else export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-fp32-variant --no-full-sampler --steps 60 > gpurun_out/${tag}_${v}_$i.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/${tag}_${v}_$i.json')); print('$v', $i, 'ms_per_step', round(d['ms_per_step'],4), 'eager', round(d['eager_loop']['ms_per_step'],4))"
done; done
