#!/bin/bash
# GPU box: training / general-shape tests, then the training step (bench.py --config c5) with the ConvNext block of the forward as a
# LayerNorm launch + two products (ARREAU_TRAIN_FUSED_MLP=0) against the sampling kernel with saves (default), alternating.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_general_shape.py -x -q > gpurun_out/abfm_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/abfm_pytest.log; [ $rc -eq 0 ] || { grep -n "^E  \|Error" gpurun_out/abfm_pytest.log | cut -c1-220 | head -n 20; exit $rc; }
for i in 1 2 3; do for v in 0 1; do
  ARREAU_TRAIN_FUSED_MLP=$v timeout -k 10 300 python3 bench.py --config c5 --no-cpu-baseline --steps 60 > gpurun_out/abfm_${v}_$i.json 2> gpurun_out/abfm_${v}_$i.err || { tail -n 20 gpurun_out/abfm_${v}_$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/abfm_${v}_$i.json')); print('fused_mlp=$v', $i, 'ms_per_step', round(d['ms_per_step'],4), 'fwd+bwd', round(d['forward_backward_ms'],4), 'loss', d['last_loss'])"
done; done
