#!/bin/bash
# usage: tools/pmc.sh <tag> "<CTR CTR ...>" ["<CTR ...>" ...]   (GPU box; one rocprofv3 --pmc pass per quoted group)
# Prints the per-launch mean of every counter for the two matrix kernels.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_$i -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph-loop --no-full-sampler $PMC_BENCH_ARGS > gpurun_out/pmc_bench_$i.json 2> gpurun_out/pmc_bench_$i.err || exit 1
  i=$((i+1))
done
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if "edge" in k or "mlp" in k or "conv" in k:
        print(k)
        for c,v in sorted(acc[k].items()):
            print("   %-28s %14.0f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
