#!/bin/bash
# usage (GPU box): tools/exp/mall_slices.sh -- does running the step slice by slice on ONE stream (each slice's stash small enough to stay
# in the 256 MiB Infinity Cache between the edge kernel that writes it and the L message kernels that read it) shorten the step?
# ARREAU_SLICE_EAGER=serial: the slices' range launches one after another on the caller's stream; ARREAU_GROUP_WGS=256: no workgroup cap.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for g in 0 2 3 4 6; do
  if [ $g = 0 ]; then unset ARREAU_SLICE_EAGER ARREAU_GROUP_WGS; gg=""; else export ARREAU_SLICE_EAGER=serial ARREAU_GROUP_WGS=256; gg="--groups $g"; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-fp32-variant --no-full-sampler --steps 40 $gg > gpurun_out/mall_g${g}_$rep.json 2> gpurun_out/mall_g${g}_$rep.err || { tail -n 20 gpurun_out/mall_g${g}_$rep.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/mall_g${g}_$rep.json')); print('groups $g rep $rep: graph ms_per_step', round(d['ms_per_step'],4), 'eager', round(d['eager_loop']['ms_per_step'],4), 'conv_proj us', round(1e3*d['roofline']['avg_launch_ms'],1))"
done; done
unset ARREAU_SLICE_EAGER ARREAU_GROUP_WGS
for g in 0 3; do
  if [ $g = 0 ]; then gg=""; else export ARREAU_SLICE_EAGER=serial ARREAU_GROUP_WGS=256; gg="--groups $g"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mall_prof_g$g -- python3 bench.py --no-cpu-baseline --no-fp32-variant --no-full-sampler --no-graph-loop --steps 30 $gg > gpurun_out/mall_prof_g$g.json 2> gpurun_out/mall_prof_g$g.err || { tail -n 20 gpurun_out/mall_prof_g$g.err; exit 1; }
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/mall_prof_g$g/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("g=$g %-60s calls %6s avg %9.1f us total %9.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
