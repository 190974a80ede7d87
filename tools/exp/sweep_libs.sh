#!/bin/bash
# usage (GPU box): tools/exp/sweep_libs.sh "<tags>" [bench args] -- per-kernel average durations (rocprofv3) of the C2 step for the
# in-tree library ("cur") and for tools/exp/ab/lib_<tag>.so of every tag (timing experiments; results of -DARREAU_EXP builds are wrong on purpose)
tags="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in cur $tags cur; do
  if [ $tag = cur ]; then unset ARREAU_HIP_LIB; else export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/lib_$tag.so; fi
  rm -rf gpurun_out/sweep_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_$tag -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-variant "$@" > gpurun_out/sweep_$tag.json 2> gpurun_out/sweep_$tag.err
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/sweep_$tag/*/*kernel_stats.csv")
if not f: print("%-8s no stats" % "$tag")
else:
    rows={r["Name"].split("(")[0][-34:]:float(r["AverageNs"])/1e3 for r in csv.DictReader(open(f[0]))}
    print("%-8s " % "$tag" + " | ".join("%s %.1f" % (k[:20], v) for k, v in list(rows.items())[:4]))
PY
done
