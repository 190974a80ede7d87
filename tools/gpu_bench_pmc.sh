#!/bin/bash
# usage (GPU box): tools/gpu_bench_pmc.sh <tag>  -- GPU tests, bench at C2 and C1, kernel stats at C2, PMC counters of the matrix kernels
tag=$1
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 4 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
for c in c2 c1; do
  timeout -k 10 300 python bench.py --config $c --steps $([ $c = c1 ] && echo 99 || echo 30) --no-cpu-baseline > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || { tail -n 30 gpurun_out/${tag}_bench_$c.err; exit 1; }
done
python - <<PY
import json
for c in ("c2", "c1"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    print(c, "ms_per_step", round(d["ms_per_step"], 4), "graph", d.get("graph_loop"), "edge_ms", round(d["roofline"]["avg_launch_ms"], 4), "E/N", d["config"]["edges_per_atom_end"])
PY
tools/prof_bench.sh ${tag} --no-fp32-variant || exit 1
cp gpurun_out/prof_${tag}/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats_c2.csv
PMC_BENCH_ARGS="--no-fp32-variant" tools/pmc.sh ${tag} "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" > gpurun_out/${tag}_pmc.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_pmc.txt; exit 1; }
cat gpurun_out/${tag}_pmc.txt
