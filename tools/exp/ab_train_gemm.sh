#!/bin/bash
# GPU box: the training step (bench.py --config c5) with the backward products on bf16x6 (default) against fp16x3
# (ARREAU_TRAIN_GEMM=fp16, unscaled: a timing bound for a scaled fp16x3 backward), alternating on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in split fp16; do
  ARREAU_TRAIN_GEMM=$v timeout -k 10 300 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/abtg_${v}_$i.json 2> gpurun_out/abtg_${v}_$i.err || { tail -n 20 gpurun_out/abtg_${v}_$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/abtg_${v}_$i.json')); print('$v', $i, 'ms_per_step', round(d['ms_per_step'],4))"
done; done
