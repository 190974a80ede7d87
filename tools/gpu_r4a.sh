#!/bin/bash
# round 4, first GPU call: the GPU suite with the new tests (bench-size loop parity, plain tolerance, training gradients at 64
# crystals / bitwise repeatability / the make-train preset), the multi-stream experiment opted in (its own log), smoke, one
# default bench line, and the timing-only sweep of conv_proj variants (tools/exp/ab/lib_*.so: wrong numbers on purpose).
tag=${1:-r04a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log
grep -h "^\[plain 1e-5\]\|^\[gradients" gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/${tag}_pytest.log | tail -n 30; exit $rc; }
ARREAU_TEST_MULTISTREAM=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k multi_stream_experiment > gpurun_out/${tag}_multistream.log 2>&1
echo "multistream rc=$?"; grep -h "multi-stream experiment\|passed\|failed" gpurun_out/${tag}_multistream.log | tail -n 3
timeout -k 10 60 tools/exp/_bin/fp8_mfma_check 2>&1 | tee gpurun_out/${tag}_fp8_mfma_check.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 || exit 1
timeout -k 10 500 python3 bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || { tail -n 30 gpurun_out/${tag}_bench_c2.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_c2.json"))
r = d["roofline"]
print("c2 value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "eager", d["eager_loop"]["ms_per_step"], "conv_proj ms", r["avg_launch_ms"], "mfma frac", round(r["frac"], 4),
      "hbm frac", round(r["hbm"]["frac"], 4), "step", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r["step"].items() if k != "note"}, "full", d["full_sampler_measured"])
PY
export ARREAU_BENCH_TIMING_ONLY=1
tools/exp/sweep_libs.sh "mfma23 bytes23 both23" --no-full-sampler
