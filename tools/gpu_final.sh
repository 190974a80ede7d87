#!/bin/bash
# usage (GPU box): tools/gpu_final.sh <tag> [benches|profiles]  -- everything a round's record needs (two calls when one would exceed a box's limit):
#   benches:  GPU suite (+ the opt-in multi-stream experiment, its own log), smoke, bench.py as the driver runs it (c2), the other BASELINE
#             configurations, the training step with exact and with split-precision forward products
#   profiles: rocprofv3 kernel statistics at c2 and of the training step, its launch list, HBM-traffic PMC passes at c2 / c4 / c5, the
#             parity report (default and heavy-tailed weights)
tag=$1; part=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $part != profiles ]; then
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log
grep -h "^\[plain 1e-5\]\|^\[gradients\|^\[fp8 cross\|^\[lone crystal\|^\[basis stash\|^\[bench path\|^   \[fp64 ref\]\|^\[loss" gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/${tag}_pytest.log | tail -n 30; exit $rc; }
ARREAU_TEST_MULTISTREAM=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k multi_stream_experiment > gpurun_out/${tag}_multistream.log 2>&1
echo "multistream rc=$?"; grep -h "multi-stream experiment\|passed\|failed" gpurun_out/${tag}_multistream.log | tail -n 3
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 || exit 1
# (the default run = what the driver times: c2 + cpu_baseline + the short legs of c1 / c4 / c5 as `other_configs`)
timeout -k 10 500 python3 bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || { tail -n 30 gpurun_out/${tag}_bench_c2.err; exit 1; }
for c in c1 c3 c4 c5; do
  steps=30; [ $c = c1 ] && steps=99; [ $c = c4 ] && steps=10
  timeout -k 10 500 python3 bench.py --config $c --steps $steps > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || { tail -n 30 gpurun_out/${tag}_bench_$c.err; exit 1; }
done
timeout -k 10 500 python3 bench.py --config c5 --hidden-dim 200 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_hidden200.json 2> gpurun_out/${tag}_bench_c5_hidden200.err || { tail -n 30 gpurun_out/${tag}_bench_c5_hidden200.err; exit 1; }
ARREAU_TRAIN_GEMM=exact timeout -k 10 500 python3 bench.py --config c5 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_exact_gemm.json 2> gpurun_out/${tag}_bench_c5_exact_gemm.err || { tail -n 30 gpurun_out/${tag}_bench_c5_exact_gemm.err; exit 1; }
python3 - <<PY
import json
for c in ("c2", "c1", "c3", "c4", "c5", "c5_hidden200", "c5_exact_gemm"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    r = d["roofline"]
    print(c, "value", round(d["value"], 1), "ms_per_step", round(d["ms_per_step"], 4), d.get("loop_mode", ""), "eager", (d.get("eager_loop") or {}).get("ms_per_step"),
          "frac", round(r["frac"], 4), "launch ms", r.get("avg_launch_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "full", (d.get("full_sampler_measured") or {}).get("crystals_per_min"),
          "step", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (r.get("step") or {}).items() if k.startswith("achieved")})
PY
fi
[ $part = benches ] && exit 0
tools/prof_bench.sh ${tag}_c2 --no-fp32-variant || exit 1
tools/gpu_prof_c5.sh ${tag}_c5 > gpurun_out/${tag}_c5_prof_tail.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_c5_prof_tail.txt; exit 1; }
tail -n 12 gpurun_out/${tag}_c5_prof_tail.txt
tools/gpu_c5_trace.sh ${tag}_c5 > gpurun_out/${tag}_c5_trace_tail.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_c5_trace_tail.txt; exit 1; }
head -n 1 gpurun_out/${tag}_c5_step.txt gpurun_out/${tag}_c5_step_full.txt
for c in c2 c4; do
  tools/hbm_traffic.sh $c > gpurun_out/${tag}_hbm_$c.txt 2>&1 || { tail -n 20 gpurun_out/${tag}_hbm_$c.txt; exit 1; }
  cat gpurun_out/${tag}_hbm_$c.txt
  cp gpurun_out/hbm_traffic_pmc_$c.json gpurun_out/${tag}_hbm_traffic_pmc_$c.json
done
tools/hbm_traffic_c5.sh > gpurun_out/${tag}_hbm_c5.txt 2>&1 || { tail -n 5 gpurun_out/${tag}_hbm_c5.txt; exit 1; }
head -n 8 gpurun_out/${tag}_hbm_c5.txt
cp gpurun_out/hbm_traffic_pmc_c5.json gpurun_out/${tag}_hbm_traffic_pmc_c5.json
timeout -k 10 900 python3 tools/parity_report.py --out gpurun_out/${tag}_parity.json > gpurun_out/${tag}_parity.log 2>&1 || { tail -n 20 gpurun_out/${tag}_parity.log; exit 1; }
tail -n 8 gpurun_out/${tag}_parity.log
timeout -k 10 900 python3 tools/parity_report.py --heavy-tailed --out gpurun_out/${tag}_parity_heavy_tailed.json > gpurun_out/${tag}_parity_heavy.log 2>&1 || { tail -n 20 gpurun_out/${tag}_parity_heavy.log; exit 1; }
tail -n 3 gpurun_out/${tag}_parity_heavy.log
python3 -c "
import json; d = json.load(open('gpurun_out/${tag}_parity.json')); print('ratios vs fp32-MFMA kernels (distance to fp64):', d['ratio_fp16x3_over_fp32mfma_vs_f64'], d['ratio_basis_form_fp8_cross_over_fp32mfma_vs_f64'])"
