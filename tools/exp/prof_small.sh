#!/bin/bash
# usage (GPU box): tools/exp/prof_small.sh <tag> [bench args]: kernel statistics of one bench run, every kernel listed
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph-loop --no-full-sampler "$@" > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err || exit 1
python3 - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:44].ljust(44), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
d=json.load(open("gpurun_out/bench_prof_$tag.json")); print("ms_per_step", d["ms_per_step"])
PY
