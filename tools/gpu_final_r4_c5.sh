#!/bin/bash
# usage (GPU box): tools/gpu_final_r4_c5.sh <tag> -- the training-step part of the record (suite, c5 benches, kernel statistics, launch lists, GEMM table)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log
grep -h "^\[gradients\|^\[loss\|^\[sgemm" gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/${tag}_pytest.log | tail -n 30; exit $rc; }
timeout -k 10 500 python3 bench.py --config c5 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err || { tail -n 30 gpurun_out/${tag}_bench_c5.err; exit 1; }
timeout -k 10 500 python3 bench.py --config c5 --hidden-dim 200 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_hidden200.json 2> gpurun_out/${tag}_bench_c5_hidden200.err || exit 1
ARREAU_TRAIN_GEMM=exact timeout -k 10 500 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_exact_gemm.json 2> gpurun_out/${tag}_bench_c5_exact_gemm.err || exit 1
python3 - <<PY
import json
for c in ("c5", "c5_hidden200", "c5_exact_gemm"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    print(c, "value", round(d["value"], 1), "ms_per_step", round(d["ms_per_step"], 4), "fb", round(d["forward_backward_ms"], 4), "frac", round(d["roofline"]["frac"], 4),
          "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
tools/gpu_prof_c5.sh ${tag}_c5 > gpurun_out/${tag}_c5_prof_tail.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_c5_prof_tail.txt; exit 1; }
tail -n 4 gpurun_out/${tag}_c5_prof_tail.txt
tools/gpu_c5_trace.sh ${tag}_c5 > gpurun_out/${tag}_c5_trace_tail.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_c5_trace_tail.txt; exit 1; }
head -n 1 gpurun_out/${tag}_c5_step.txt gpurun_out/${tag}_c5_step_full.txt
grep "kernel time" gpurun_out/${tag}_c5_step.txt gpurun_out/${tag}_c5_step_full.txt
timeout -k 10 300 tools/exp/_bin/sgemm_bench > gpurun_out/${tag}_sgemm_bench.txt 2>&1 || { tail -n 5 gpurun_out/${tag}_sgemm_bench.txt; exit 1; }
cut -c1-37,62-260 gpurun_out/${tag}_sgemm_bench.txt | head -n 12
tools/hbm_traffic_c5.sh > gpurun_out/${tag}_hbm_c5.txt 2>&1 || { tail -n 5 gpurun_out/${tag}_hbm_c5.txt; exit 1; }
head -n 8 gpurun_out/${tag}_hbm_c5.txt
cp gpurun_out/hbm_traffic_pmc_c5.json gpurun_out/${tag}_hbm_traffic_pmc_c5.json
timeout -k 10 500 python3 bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || { tail -n 30 gpurun_out/${tag}_bench_c2.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_c2.json')); print('c2', round(d['value'],1), round(d['ms_per_step'],4), d['roofline'].get('avg_launch_ms'), (d.get('full_sampler_measured') or {}).get('crystals_per_min'))"
