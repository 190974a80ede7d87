#!/bin/bash
# usage (GPU box): tools/gpu_train_check.sh <tag>  -- training + general-shape GPU tests, then kernel stats of the training bench (c5)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_general_shape.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 5 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c5 -- python3 bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof_c5.json 2> gpurun_out/${tag}_prof_c5.err || { tail -n 20 gpurun_out/${tag}_prof_c5.err; exit 1; }
python3 - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/prof_${tag}_c5/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(" ", r["Name"][:48].ljust(48), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
d=json.load(open("gpurun_out/${tag}_prof_c5.json")); print("ms_per_step", d["ms_per_step"], "fb", d.get("forward_backward_ms"), "value", d["value"])
PY
