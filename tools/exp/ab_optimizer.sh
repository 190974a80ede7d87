#!/bin/bash
# GPU box: the training tests, then the training step (bench.py --config c5) with torch's clip + fused Adam (ARREAU_TORCH_ADAM=1) against
# the library's two launches (arreau_amd/optim.py), alternating on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_general_shape.py -x -q > gpurun_out/abopt_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/abopt_pytest.log; [ $rc -eq 0 ] || exit $rc
for i in 1 2; do for v in 1 0; do
  ARREAU_TORCH_ADAM=$v timeout -k 10 300 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/abopt_${v}_$i.json 2> gpurun_out/abopt_${v}_$i.err || { tail -n 20 gpurun_out/abopt_${v}_$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/abopt_${v}_$i.json')); print('torch_adam=$v', $i, 'ms_per_step', round(d['ms_per_step'],4), 'fwd+bwd', round(d['forward_backward_ms'],4), 'loss', d['last_loss'])"
done; done
