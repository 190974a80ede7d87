#!/bin/bash
# usage (GPU box): tools/gpu_staircase.sh <tag>  -- kernel durations against batch size for both MLP tile geometries
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for nb in 1 2; do
  ARREAU_MLP_NB=$nb timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stair_${tag}_nb$nb -- python3 tools/exp/mlp_staircase.py > gpurun_out/${tag}_stair_nb$nb.log 2>&1 || { tail -n 20 gpurun_out/${tag}_stair_nb$nb.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
for nb in (1,2):
    f=glob.glob("gpurun_out/stair_${tag}_nb%d/*/*kernel_trace.csv" % nb)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for key in ("mlp_kernel_f16x3_m16","conv_kernel_streamed","edge_kernel_f16x3"):
            if key in n:
                agg[(key,int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    print("NB", nb)
    for k in sorted(agg):
        v=sorted(agg[k]); print("  ", k[0][:22], "workgroups", k[1], "median us %.1f" % v[len(v)//2], "n", len(v))
PY
