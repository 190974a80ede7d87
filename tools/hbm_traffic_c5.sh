#!/bin/bash
# usage (GPU box): tools/hbm_traffic_c5.sh -- HBM bytes of ONE training step (bench.py --config c5): two rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes; gfx950 correction: FETCH_SIZE counts wide coalesced reads at 1/2), the
# dispatches of one forward + backward step and of one step with its optimizer part cut out by their `prep_kernel` launches
# -> gpurun_out/hbm_traffic_pmc_c5.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (per-kernel means and counters of the WORKLOAD: without the three small evaluations of arreau_model_create's calibration batch, which
# launch the same kernels on 320 atoms; the synthetic checkpoint keeps both fp8 formats either way)
export ARREAU_CALIBRATE=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_traffic_c5_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_traffic_c5_$c -- python3 bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_traffic_c5_$c.json 2> gpurun_out/pmc_traffic_c5_$c.err || { tail -n 5 gpurun_out/pmc_traffic_c5_$c.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, collections, re
tot = {}
top = collections.defaultdict(lambda: [0.0, 0.0])
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = glob.glob("gpurun_out/pmc_traffic_c5_%s/*/*counter_collection.csv" % c)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "prep_kernel" in r["Kernel_Name"]]
    for tag, (a, b) in (("forward_backward", (marks[-3], marks[-2])), ("with_optimizer", (marks[4], marks[5]))):
        tot[(tag, c)] = sum(float(r["Counter_Value"]) for r in rows[a:b])
        tot[(tag, "launches")] = b - a
    for r in rows[marks[-3]:marks[-2]]:
        top[re.sub(r"\(anonymous namespace\)::|arreau_sgemm_detail::|void ", "", r["Kernel_Name"]).split("(")[0][:70]][ci] += float(r["Counter_Value"])
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/hbm_traffic_c5.sh) over `python bench.py --config c5 --steps 4 "
               "--warmup 2 --no-cpu-baseline`; the dispatches of one step cut out by their prep_kernel launches; KB summed over the step; "
               "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled",
       "config": "c5", "crystals_per_gpu": 64, "atoms_per_crystal": 0}
for tag in ("forward_backward", "with_optimizer"):
    out["hbm_bytes_per_step_" + tag] = (2 * tot[(tag, "FETCH_SIZE")] + tot[(tag, "WRITE_SIZE")]) * 1024
    out["launches_" + tag] = tot[(tag, "launches")]
out["largest_kernels_forward_backward_bytes"] = {k: (2 * v[0] + v[1]) * 1024 for k, v in sorted(top.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:24]}
json.dump(out, open("gpurun_out/hbm_traffic_pmc_c5.json", "w"), indent=1)
print("training step: %.3f GB forward + backward (%d launches), %.3f GB with the optimizer part" % (
    out["hbm_bytes_per_step_forward_backward"] / 1e9, out["launches_forward_backward"], out["hbm_bytes_per_step_with_optimizer"] / 1e9))
for k, v in list(out["largest_kernels_forward_backward_bytes"].items())[:24]:
    print("  %-70s %8.1f MB" % (k, v / 1e6))
PY
