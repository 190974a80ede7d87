#!/bin/bash
# usage (GPU box): tools/gpu_prof_c5_h.sh <tag> <hidden_dim>  -- kernel stats of the training bench at a given hidden_dim
tag=$1; hd=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c5h -- python3 bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline --hidden-dim $hd > gpurun_out/${tag}_prof_c5h.json 2> gpurun_out/${tag}_prof_c5h.err || { tail -n 20 gpurun_out/${tag}_prof_c5h.err; exit 1; }
python3 - <<PY
import csv,glob,json
f=sorted(glob.glob("gpurun_out/prof_${tag}_c5h/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:22]:
    print(" ", r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
d=json.load(open("gpurun_out/${tag}_prof_c5h.json")); print("ms_per_step", d["ms_per_step"], "fb", d.get("forward_backward_ms"), "value", d["value"])
PY
