#!/bin/bash
# usage: tools/ab.sh [bench args]  (GPU box): time the in-tree library against another build of it (tools/exp/ab/${AB_LIB:-lib_prev.so}),
# alternating, on the same box (box-to-box spread is +-3 %, within one box +-0.5 %).  Prepare the other build in this container first:
#   git stash; python -m arreau_amd.build; mkdir -p tools/exp/ab; cp arreau_amd/csrc/libarreau_hip.so tools/exp/ab/lib_prev.so
#   git stash pop; python -m arreau_amd.build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for which in prev cur; do
    if [ $which = prev ]; then export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/${AB_LIB:-lib_prev.so}; else unset ARREAU_HIP_LIB; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_${which}_$rep -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/ab_${which}_$rep.json 2> gpurun_out/ab_${which}_$rep.err || exit 1
    python - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/ab_${which}_$rep/*/*kernel_stats.csv")[0]
rows={r["Name"].split("(")[0][-32:]:float(r["AverageNs"])/1e3 for r in csv.DictReader(open(f))}
d=json.load(open("gpurun_out/ab_${which}_$rep.json"))
print("%-5s rep $rep: step %.3f ms | " % ("$which", d["ms_per_step"]) + " | ".join("%s %.1f" % (k[:18], v) for k, v in list(rows.items())[:4]))
PY
  done
done
