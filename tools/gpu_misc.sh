#!/bin/bash
# usage (GPU box): tools/gpu_misc.sh <tag> -- GPU suite; self-launched 2-rank rehearsal of bench.py on one device (sampling + training); c2 / c1 bench
tag=$1
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 4 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
ARREAU_BENCH_BACKEND=gloo ARREAU_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/${tag}_bench_2rank.json 2> gpurun_out/${tag}_bench_2rank.err || { tail -n 30 gpurun_out/${tag}_bench_2rank.err; exit 1; }
ARREAU_BENCH_BACKEND=gloo ARREAU_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --config c5 --steps 5 --warmup 3 > gpurun_out/${tag}_bench_2rank_c5.json 2> gpurun_out/${tag}_bench_2rank_c5.err || { tail -n 30 gpurun_out/${tag}_bench_2rank_c5.err; exit 1; }
for c in c2 c1; do
  timeout -k 10 300 python bench.py --config $c --steps $([ $c = c1 ] && echo 99 || echo 30) --no-cpu-baseline > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || { tail -n 30 gpurun_out/${tag}_bench_$c.err; exit 1; }
done
python - <<PY
import json
for f in ("2rank", "2rank_c5", "c2", "c1"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % f))
    print(f, "n_gpus", d["n_gpus"], "ms_per_step", round(d["ms_per_step"], 4), "per_rank_ms", [round(x, 3) for x in d["per_rank_ms"]], "value", round(d["value"], 1))
PY
tools/prof_bench.sh ${tag} --no-fp32-variant | head -n 12
