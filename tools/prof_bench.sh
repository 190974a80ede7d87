#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]   (GPU box): kernel stats of one bench run, no tests
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (per-kernel means and counters of the WORKLOAD: without the three small evaluations of arreau_model_create's calibration batch, which
# launch the same kernels on 320 atoms; the synthetic checkpoint keeps both fp8 formats either way)
export ARREAU_CALIBRATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph-loop --no-full-sampler "$@" > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err || exit 1
python - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:36].ljust(36), r["Calls"].rjust(4), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
d=json.load(open("gpurun_out/bench_prof_$tag.json")); print("ms_per_step", d["ms_per_step"], "edge_ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
PY
