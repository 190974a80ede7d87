#!/bin/bash
# usage (GPU box): tools/gpu_prof_c5.sh <tag> [bench args] -- rocprofv3 kernel statistics of the training step (bench.py --config c5)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (per-kernel means and counters of the WORKLOAD: without the three small evaluations of arreau_model_create's calibration batch, which
# launch the same kernels on 320 atoms; the synthetic checkpoint keeps both fp8 formats either way)
export ARREAU_CALIBRATE=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/${tag}_prof.json 2> gpurun_out/${tag}_prof.err || { tail -n 20 gpurun_out/${tag}_prof.err; exit 1; }
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 36.0
print("kernel time per step (35 steps + warm-up ~ 36): %.3f ms, launches per step %.0f" % (tot / 1e6 / steps, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:26]:
    print(r["Name"][:84].ljust(84), ("%.1f" % (int(r["Calls"]) / steps)).rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), "us", ("%.3f" % (float(r["TotalDurationNs"]) / 1e6 / steps)).rjust(7), "ms/step")
PY
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_prof.json')); print('ms_per_step', d['ms_per_step'], d['roofline']['frac'])"
