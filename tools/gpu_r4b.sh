#!/bin/bash
# round 4: training tests + GEMM micro-benchmark (exact / fp16x3 / bf16x6) + c5 bench with exact and split products
tag=${1:-r04b}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 tools/exp/_bin/sgemm_bench > gpurun_out/${tag}_sgemm_bench.txt 2>&1; echo "sgemm_bench rc=$?"; cat gpurun_out/${tag}_sgemm_bench.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_general_shape.py -m gpu -x -q -s > gpurun_out/${tag}_pytest_train.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest_train.log; grep -h "^\[gradients\|^\[loss" gpurun_out/${tag}_pytest_train.log
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/${tag}_pytest_train.log | tail -n 30; }
for mode in exact split; do
  ARREAU_TRAIN_GEMM=$mode timeout -k 10 400 python3 bench.py --config c5 --steps 30 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_$mode.json 2> gpurun_out/${tag}_bench_c5_$mode.err || { tail -n 20 gpurun_out/${tag}_bench_c5_$mode.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/${tag}_bench_c5_$mode.json')); print('c5 $mode ms_per_step', round(d['ms_per_step'],4), 'fb_ms', round(d['forward_backward_ms'],4), 'loss', d['last_loss'])"
done
