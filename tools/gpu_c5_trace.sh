#!/bin/bash
# usage (GPU box): tools/gpu_c5_trace.sh <tag> -- launch list of one training step (rocprofv3 kernel trace)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python3 bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/${tag}_trace.json 2> gpurun_out/${tag}_trace.err || { tail -n 20 gpurun_out/${tag}_trace.err; exit 1; }
python3 tools/exp/c5_step_trace.py gpurun_out/trace_$tag > gpurun_out/${tag}_step.txt && python3 tools/exp/c5_step_trace.py gpurun_out/trace_$tag full > gpurun_out/${tag}_step_full.txt && tail -n 45 gpurun_out/${tag}_step.txt
find gpurun_out/trace_$tag -name "*.csv" -size +20M -delete
