#!/bin/bash
# usage (GPU box): tools/exp/sweep_scores.sh "<tags>" -- per-kernel average durations of four score evaluations of ONE valid
# 256 x 20 state (tools/determinism.py) for the in-tree library and tools/exp/ab/lib_<tag>.so: for timing-only builds of the
# edge kernel, whose wrong outputs would change the state (and the edge count) of a sampling loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in cur $1 cur; do
  if [ $tag = cur ]; then unset ARREAU_HIP_LIB; else export ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/lib_$tag.so; fi
  rm -rf gpurun_out/sws_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sws_$tag -- python tools/determinism.py 256 20 > gpurun_out/sws_$tag.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/sws_$tag/*/*kernel_stats.csv")
rows={r["Name"].split("(")[0][-30:]:(float(r["MinNs"])/1e3, r["Calls"]) for r in csv.DictReader(open(f[0]))} if f else {}
print("%-6s " % "$tag" + " | ".join("%s min %.1f (%s)" % (k[:18], v[0], v[1]) for k, v in list(rows.items())[:3]))
PY
done
