#!/bin/bash
# usage (GPU box): tools/gpu_edge_split_sweep.sh <tag>  -- edge / read-out kernel time against batch size, persistent vs small-launch form
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for sp in 0 1; do
  SIZES=3,6,10,13,19,26,38,51 ARREAU_EDGE_SPLIT=$sp ARREAU_READOUT_SPLIT=$sp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/esweep_${tag}_$sp -- python3 tools/exp/mlp_staircase.py > gpurun_out/${tag}_esweep_$sp.log 2>&1 || { tail -n 20 gpurun_out/${tag}_esweep_$sp.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
for sp in (0,1):
    f=sorted(glob.glob("gpurun_out/esweep_${tag}_%d/*/*kernel_trace.csv" % sp))[-1]
    rows=list(csv.DictReader(open(f)))
    # group launches by order: sizes in order, 3 evaluations each
    for key in ("edge_kernel_f16x3", "readout_mfma_kernel"):
        d=[(int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])) for r in rows if key in r["Kernel_Name"]]
        d.sort()
        out=[]
        for i in range(0, len(d), 3):
            v=sorted(x[1] for x in d[i:i+3]); out.append((d[i][2], d[i][3], round(v[1],1)))
        print("split" if sp else "persistent", key, out)
PY
