#!/bin/bash
# usage (GPU box): tools/exp/prof_config.sh <config> [bench args] -- rocprofv3 kernel statistics of one bench configuration (eager loop)
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profc_$cfg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profc_$cfg -- python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-graph-loop --no-full-sampler --no-fp32-variant "$@" > gpurun_out/profc_$cfg.json 2> gpurun_out/profc_$cfg.err || { tail -n 5 gpurun_out/profc_$cfg.err; exit 1; }
python - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/profc_$cfg/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("$cfg", r["Name"][:40].ljust(40), r["Calls"].rjust(5), "%10.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
d=json.load(open("gpurun_out/profc_$cfg.json")); print("$cfg ms_per_step", d["ms_per_step"])
PY
