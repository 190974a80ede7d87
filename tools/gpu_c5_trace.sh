#!/bin/bash
# usage (GPU box): tools/gpu_c5_trace.sh <tag> -- launch list of one training step (rocprofv3 kernel trace)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (per-kernel means and counters of the WORKLOAD: without the three small evaluations of arreau_model_create's calibration batch, which
# launch the same kernels on 320 atoms; the synthetic checkpoint keeps both fp8 formats either way)
export ARREAU_CALIBRATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python3 bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/${tag}_trace.json 2> gpurun_out/${tag}_trace.err || { tail -n 20 gpurun_out/${tag}_trace.err; exit 1; }
python3 tools/exp/c5_step_trace.py gpurun_out/trace_$tag > gpurun_out/${tag}_step.txt && python3 tools/exp/c5_step_trace.py gpurun_out/trace_$tag full > gpurun_out/${tag}_step_full.txt && tail -n 45 gpurun_out/${tag}_step.txt
find gpurun_out/trace_$tag -name "*.csv" -size +20M -delete
