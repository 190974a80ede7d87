#!/bin/bash
# usage (GPU box): tools/gpu_prof_c1.sh <tag>  -- kernel stats of the 1 x 8 sampler bench (c1)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c1 -- python3 bench.py --config c1 --steps 99 --no-cpu-baseline --no-graph-loop --no-full-sampler --no-fp32-variant > gpurun_out/${tag}_prof_c1.json 2> gpurun_out/${tag}_prof_c1.err || { tail -n 20 gpurun_out/${tag}_prof_c1.err; exit 1; }
python3 - <<PY
import csv,glob,json
f=sorted(glob.glob("gpurun_out/prof_${tag}_c1/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(" ", r["Name"][:48].ljust(48), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
d=json.load(open("gpurun_out/${tag}_prof_c1.json")); print("ms_per_step", d["ms_per_step"])
PY
