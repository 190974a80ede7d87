#!/bin/bash
# usage (GPU box): tools/gpu_final_c5.sh <tag> -- the training step's part of a round's record alone (after a change that touches only
# train_net.hip / train.hip): its GPU tests, bench.py --config c5 (+ hidden_dim 200, exact products), kernel statistics, launch lists,
# PMC traffic.  tools/collect_profiles_c5.sh <tag> copies the results into profiles/.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_general_shape.py tests/test_gpu_sgemm.py -m gpu -x -q -s > gpurun_out/${tag}_pytest_training.log 2>&1; rc=$?
tail -n 2 gpurun_out/${tag}_pytest_training.log
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/${tag}_pytest_training.log | tail -n 20; exit $rc; }
timeout -k 10 500 python3 bench.py --config c5 --steps 30 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err || { tail -n 30 gpurun_out/${tag}_bench_c5.err; exit 1; }
timeout -k 10 500 python3 bench.py --config c5 --hidden-dim 200 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_hidden200.json 2> gpurun_out/${tag}_bench_c5_hidden200.err || exit 1
ARREAU_TRAIN_GEMM=exact timeout -k 10 500 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_exact_gemm.json 2> gpurun_out/${tag}_bench_c5_exact_gemm.err || exit 1
ARREAU_TRAIN_FUSE=0 timeout -k 10 500 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_one_kernel_per_operation.json 2> gpurun_out/${tag}_bench_c5_plain.err || exit 1
python3 - <<PY
import json
for c in ("c5", "c5_hidden200", "c5_exact_gemm", "c5_one_kernel_per_operation"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    print(c, "ms_per_step", round(d["ms_per_step"], 4), "forward+backward", round(d["forward_backward_ms"], 4), "frac", round(d["roofline"]["frac"], 4))
PY
tools/gpu_prof_c5.sh ${tag}_c5 > gpurun_out/${tag}_c5_prof_tail.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_c5_prof_tail.txt; exit 1; }
tools/gpu_c5_trace.sh ${tag}_c5 > gpurun_out/${tag}_c5_trace_tail.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_c5_trace_tail.txt; exit 1; }
head -n 1 gpurun_out/${tag}_c5_step.txt gpurun_out/${tag}_c5_step_full.txt
tools/hbm_traffic_c5.sh > gpurun_out/${tag}_hbm_c5.txt 2>&1 || { tail -n 5 gpurun_out/${tag}_hbm_c5.txt; exit 1; }
head -n 4 gpurun_out/${tag}_hbm_c5.txt
cp gpurun_out/hbm_traffic_pmc_c5.json gpurun_out/${tag}_hbm_traffic_pmc_c5.json
