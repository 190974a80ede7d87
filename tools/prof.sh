#!/bin/bash
# usage: tools_prof.sh <tag> [pytest -k filter]   (runs on the GPU box via gpurun)
tag=$1; filt=${2:-"predict_scores or full_size or trajectory or forward"}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$filt" > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$?; tail -1 gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
python - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:36].ljust(36), r["Calls"].rjust(4), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
d=json.load(open("gpurun_out/bench_prof.json")); print("ms_per_step", d["ms_per_step"], "edge_ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
PY
