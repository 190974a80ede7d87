#!/bin/bash
# round 4: quick check of a conv_proj change -- the tests that exercise the basis form, then x16 / x8 alternating
tag=${1:-r04d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "batch_independence or launch_geometry or loop_at_the_benchmark or full_size_architecture or plain_tolerance or range_launches or basis_stash or many_ragged or counted_waits or multi_stream or sliced" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log
grep -h "^\[plain 1e-5\]\|^\[fp8 cross\|^\[lone crystal\|^\[basis stash" gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert \|FAILED" gpurun_out/${tag}_pytest.log | tail -n 30; exit $rc; }
for i in 1 2; do
for v in x16 x8; do
  env=1; [ $v = x16 ] && env=0
  ARREAU_CROSS_FP8=$env timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-fp32-variant --no-full-sampler --steps 60 > gpurun_out/${tag}_c2_${v}_$i.json 2> gpurun_out/${tag}_c2_${v}_$i.err || { tail -n 20 gpurun_out/${tag}_c2_${v}_$i.err; exit 1; }
done; done
python3 - <<PY
import json
for v in ("x16", "x8"):
    for i in (1, 2):
        d = json.load(open("gpurun_out/${tag}_c2_%s_%d.json" % (v, i)))
        r = d["roofline"]
        print(v, i, "ms_per_step", round(d["ms_per_step"], 4), "eager", round(d["eager_loop"]["ms_per_step"], 4), "conv_proj us", round(1e3 * r["avg_launch_ms"], 1),
              "frac", round(r["frac"], 4), "of f16x3 roof", round(r["frac_of_f16x3_roof"], 4), r["cross_products"])
PY
