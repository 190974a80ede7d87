#!/bin/bash
# usage (GPU box): tools/gpu_pipe.sh <tag> -- node-phase pipeline experiment at C2 (graph loop), alternating with the baselines
tag=$1
cd $GRAFT_REPO_ROOT
run() { # name, env, args
  env $2 timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-variant $3 > gpurun_out/${tag}_$1.json 2> gpurun_out/${tag}_$1.err || { tail -n 20 gpurun_out/${tag}_$1.err; return 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$1.json"))
print("%-14s eager %.4f ms  graph %.4f ms" % ("$1", d["ms_per_step"], d["graph_loop"]["ms_per_step"]))
PY
}
for rep in 1 2; do
  run g1_$rep "A=1" "--groups 1" || exit 1
  run g2_$rep "A=1" "--groups 2" || exit 1
  run pipe_reg_$rep "ARREAU_NODE_PIPELINE=1 ARREAU_NODE_PIPELINE_CONV=0" "--groups 2" || exit 1
  run pipe_str_$rep "ARREAU_NODE_PIPELINE=1 ARREAU_NODE_PIPELINE_CONV=1" "--groups 2" || exit 1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARREAU_NODE_PIPELINE=1 ARREAU_NODE_PIPELINE_CONV=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_pipe -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-variant --groups 2 > /dev/null 2> gpurun_out/${tag}_prof.err
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_${tag}_pipe/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:44].ljust(44), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
