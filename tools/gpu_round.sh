#!/bin/bash
# usage (GPU box): tools/gpu_round.sh <tag>   -- GPU test suite, parity report, default bench; everything logged under gpurun_out/
tag=${1:-r02}
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 15 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/parity_report.py --out gpurun_out/${tag}_parity.json > gpurun_out/${tag}_parity.log 2>&1 || { tail -n 30 gpurun_out/${tag}_parity.log; exit 1; }
tail -n 16 gpurun_out/${tag}_parity.log
timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || { tail -n 30 gpurun_out/${tag}_bench_c2.err; exit 1; }
cat gpurun_out/${tag}_bench_c2.json
