#!/bin/bash
# usage (GPU box): tools/gpu_prof_c5.sh <tag>  -- kernel stats of the training bench (c5), then the same bench at the
# reference's `make train` preset hidden_dim = 200, then kernel stats of the 1 x 8 sampler (c1)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c5 -- python3 bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof_c5.json 2> gpurun_out/${tag}_prof_c5.err || { tail -n 20 gpurun_out/${tag}_prof_c5.err; exit 1; }
timeout -k 10 400 python3 bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline --hidden-dim 200 > gpurun_out/${tag}_bench_c5_h200.json 2> gpurun_out/${tag}_bench_c5_h200.err || { tail -n 20 gpurun_out/${tag}_bench_c5_h200.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c1 -- python3 bench.py --config c1 --steps 99 --no-cpu-baseline --no-fp32-variant > gpurun_out/${tag}_prof_c1.json 2> gpurun_out/${tag}_prof_c1.err || { tail -n 20 gpurun_out/${tag}_prof_c1.err; exit 1; }
python3 - <<PY
import csv,glob,json
for c in ("c5", "c1"):
    f=glob.glob("gpurun_out/prof_${tag}_%s/*/*kernel_stats.csv" % c)[0]
    print(c)
    for r in list(csv.DictReader(open(f)))[:24]:
        print(" ", r["Name"][:48].ljust(48), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
for n in ("prof_c5", "bench_c5_h200", "prof_c1"):
    d=json.load(open("gpurun_out/${tag}_%s.json" % n)); print(n, "ms_per_step", d["ms_per_step"], "fb", d.get("forward_backward_ms"), "value", d["value"])
PY
