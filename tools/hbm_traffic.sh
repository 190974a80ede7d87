#!/bin/bash
# usage: tools/hbm_traffic.sh   (GPU box).  Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the bench workload
# -> gpurun_out/hbm_traffic_pmc.json in the layout bench.py reads from profiles/hbm_traffic_pmc.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_traffic_$c -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph-loop --no-full-sampler --no-fp32-variant > gpurun_out/pmc_traffic_$c.json 2> gpurun_out/pmc_traffic_$c.err || exit 1
done
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_traffic_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/hbm_traffic.sh), `python bench.py --steps 6 "
               "--warmup 2 --no-cpu-baseline --no-graph-loop --no-full-sampler`, C2 workload. KB per launch (mean). gfx950 correction (MI355X_MICROARCH.md, HBM): "
               "FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled in hbm_bytes_corrected.",
       "crystals_per_gpu": 256, "atoms_per_crystal": 20, "kernels": {}}
edge = None
for k, v in acc.items():
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
    w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    out["kernels"][k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": (2 * f + w) * 1024}
    if "edge_kernel_f16x3" in k:
        edge = k
out["edge_kernel"] = edge
out["edge_kernel_hbm_bytes_per_launch"] = out["kernels"][edge]["hbm_bytes_corrected"]
json.dump(out, open("gpurun_out/hbm_traffic_pmc.json", "w"), indent=1)
for k in ("edge_kernel", "conv_kernel", "mlp_kernel"):
    for n, v in out["kernels"].items():
        if k in n:
            print(n, "%.1f MB per launch" % (v["hbm_bytes_corrected"] / 1e6))
PY
