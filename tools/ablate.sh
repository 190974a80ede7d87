#!/bin/bash
# usage: tools_abl.sh VAR v1 v2 ...  : per-kernel mean times for each value of env VAR
var=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export $var=$v
  rm -rf gpurun_out/abl_tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_tmp -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/abl_tmp/*/*kernel_stats.csv")[0]
rows={r["Name"].split("(")[0][:28]:float(r["AverageNs"])/1e3 for r in csv.DictReader(open(f))}
print("$var=$v", " ".join("%s=%.1f"%(k.replace("void ",""),v) for k,v in rows.items() if any(s in k for s in ("edge","mlp","conv"))))
PY
done
