#!/bin/bash
# usage (GPU box): tools/gpu_final.sh <tag> [benches|profiles]  (two calls when one would exceed a box's time limit) -- everything a round's record needs, in one call: GPU suite + smoke, bench.py as
# the driver runs it (c2) and at the other BASELINE configurations, kernel statistics and HBM-traffic PMC passes at c2
tag=$1; part=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $part != profiles ]; then
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 || exit 1
timeout -k 10 500 python3 bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || { tail -n 30 gpurun_out/${tag}_bench_c2.err; exit 1; }
for c in c1 c3 c4 c5; do
  steps=30; [ $c = c1 ] && steps=99; [ $c = c4 ] && steps=10
  timeout -k 10 500 python3 bench.py --config $c --steps $steps > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || { tail -n 30 gpurun_out/${tag}_bench_$c.err; exit 1; }
done
timeout -k 10 500 python3 bench.py --config c5 --hidden-dim 200 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_hidden200.json 2> gpurun_out/${tag}_bench_c5_hidden200.err || { tail -n 30 gpurun_out/${tag}_bench_c5_hidden200.err; exit 1; }
python3 - <<PY
import json
for c in ("c2", "c1", "c3", "c4", "c5", "c5_hidden200"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    print(c, "value", round(d["value"], 1), "ms_per_step", round(d["ms_per_step"], 4), d.get("loop_mode", ""), "eager", (d.get("eager_loop") or {}).get("ms_per_step"),
          "frac", round(d["roofline"]["frac"], 4), "cpu", (d.get("cpu_baseline") or {}).get("value"), "full", (d.get("full_sampler_measured") or {}).get("crystals_per_min"))
PY
fi
[ $part = benches ] && exit 0
tools/prof_bench.sh ${tag}_c2 --no-fp32-variant || exit 1
for c in c2 c4; do
  tools/hbm_traffic.sh $c > gpurun_out/${tag}_hbm_$c.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_hbm_$c.txt; exit 1; }
  cat gpurun_out/${tag}_hbm_$c.txt
  cp gpurun_out/hbm_traffic_pmc_$c.json gpurun_out/${tag}_hbm_traffic_pmc_$c.json
done
timeout -k 10 600 python3 tools/parity_report.py --out gpurun_out/${tag}_parity.json > gpurun_out/${tag}_parity.log 2>&1 || { tail -n 20 gpurun_out/${tag}_parity.log; exit 1; }
tail -n 6 gpurun_out/${tag}_parity.log
