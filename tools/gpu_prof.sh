#!/bin/bash
# usage (GPU box): tools/gpu_prof.sh <tag> [bench args...] -- rocprofv3 kernel statistics of an eager bench run (whole-batch launches),
# summary table on stdout, CSV under gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (per-kernel means and counters of the WORKLOAD: without the three small evaluations of arreau_model_create's calibration batch, which
# launch the same kernels on 320 atoms; the synthetic checkpoint keeps both fp8 formats either way)
export ARREAU_CALIBRATE=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-variant --no-full-sampler --no-graph-loop "$@" > gpurun_out/${tag}_prof.json 2> gpurun_out/${tag}_prof.err || { tail -n 20 gpurun_out/${tag}_prof.err; exit 1; }
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows[:18]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), "us", r["Percentage"].rjust(6))
PY
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_prof.json')); print('ms_per_step (eager, under the profiler)', d['ms_per_step'])"
