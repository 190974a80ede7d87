#!/bin/bash
# usage (GPU box): tools/gpu_c1.sh <tag>  -- GPU tests, smoke, then bench + kernel stats at config 1 (1 crystal x 8 atoms, T=100)
tag=${1:-r02}
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 6 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3 || exit 1
timeout -k 10 300 python bench.py --config c1 --steps 200 --warmup 20 > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err || { tail -n 30 gpurun_out/${tag}_bench_c1.err; exit 1; }
cat gpurun_out/${tag}_bench_c1.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c1 -- python bench.py --config c1 --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-variant > gpurun_out/${tag}_prof_c1.json 2> gpurun_out/${tag}_prof_c1.err || { tail -n 30 gpurun_out/${tag}_prof_c1.err; exit 1; }
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_${tag}_c1/*/*kernel_stats.csv")[0]
import shutil; shutil.copy(f, "gpurun_out/${tag}_kernel_stats_c1.csv")
for r in list(csv.DictReader(open(f)))[:24]:
    print(r["Name"][:48].ljust(48), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || { tail -n 30 gpurun_out/${tag}_bench_c2.err; exit 1; }
python - <<PY
import json
for c in ("c1", "c2"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    print(c, "ms_per_step", round(d["ms_per_step"], 4), "free_loop", d.get("free_running_loop"), "edge_ms", round(d["roofline"]["avg_launch_ms"], 4))
PY
