#!/bin/bash
# usage (here, after tools/gpu_final.sh <tag> benches + profiles ran on the GPU box): tools/collect_profiles.sh <tag>
# copies the round's record from gpurun_out/ (scratch) into profiles/ (tracked)
tag=$1
for c in c1 c2 c3 c4 c5 c5_hidden200; do cp gpurun_out/${tag}_bench_$c.json profiles/${tag}_bench_$c.json; done
cp $(ls gpurun_out/prof_${tag}_c2/*/*kernel_stats.csv | head -1) profiles/${tag}_kernel_stats_bench_c2_steps30.csv
for c in c2 c4; do cp gpurun_out/${tag}_hbm_traffic_pmc_$c.json profiles/${tag}_hbm_traffic_pmc_$c.json; done
python3 - <<PY
import json
json.dump([json.load(open("profiles/${tag}_hbm_traffic_pmc_%s.json" % c)) for c in ("c2", "c4")], open("profiles/hbm_traffic_pmc.json", "w"), indent=1)
PY
cp gpurun_out/${tag}_parity.json profiles/parity_r03.json
tail -n 3 gpurun_out/${tag}_pytest.log > profiles/${tag}_pytest_tail.txt
