#!/bin/bash
# usage (GPU box): tools/gpu_full.sh <tag>  -- whole GPU suite + smoke, benches at C2 / C1 / C5, HBM traffic PMC, A/B of ARREAU_MLP_NB=1 at C2
tag=$1
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 4 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || { tail -n 30 gpurun_out/${tag}_bench_c2.err; exit 1; }
timeout -k 10 300 python bench.py --config c1 --steps 99 > gpurun_out/${tag}_bench_c1.json 2> gpurun_out/${tag}_bench_c1.err || { tail -n 30 gpurun_out/${tag}_bench_c1.err; exit 1; }
timeout -k 10 300 python bench.py --config c5 --steps 20 --warmup 3 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err || { tail -n 30 gpurun_out/${tag}_bench_c5.err; exit 1; }
python - <<PY
import json
for c in ("c2", "c1", "c5"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    print(c, "ms_per_step", round(d["ms_per_step"], 4), "graph", (d.get("graph_loop") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "fb_ms", d.get("forward_backward_ms"))
PY
tools/hbm_traffic.sh > gpurun_out/${tag}_hbm.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_hbm.txt; exit 1; }
cat gpurun_out/${tag}_hbm.txt
tools/ab_env.sh "ARREAU_MLP_NB=1" --no-fp32-variant
