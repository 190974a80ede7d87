#!/bin/bash
# usage (here, after `tools/gpu_final.sh <btag> benches` and `tools/gpu_final.sh <ptag> profiles` ran on the GPU box):
#   tools/collect_profiles.sh <btag> <ptag> <round>        e.g.  tools/collect_profiles.sh r05m r05n r05
# copies the round's record from gpurun_out/ (scratch) into profiles/ (tracked)
b=$1; p=$2; r=$3
for c in c1 c2 c3 c4 c5 c5_hidden200 c5_exact_gemm; do cp gpurun_out/${b}_bench_$c.json profiles/${b}_bench_$c.json; done
cp $(ls gpurun_out/prof_${p}_c2/*/*kernel_stats.csv | head -1) profiles/${p}_kernel_stats_bench_c2_steps30.csv
cp gpurun_out/bench_prof_${p}_c2.json profiles/${p}_bench_c2_profiled_run.json
cp gpurun_out/${p}_c5_kernel_stats.csv profiles/${p}_kernel_stats_bench_c5.csv
cp gpurun_out/${p}_c5_step.txt profiles/${p}_c5_step_launch_list_forward_backward.txt
cp gpurun_out/${p}_c5_step_full.txt profiles/${p}_c5_step_launch_list_with_optimizer.txt
for c in c2 c4 c5; do cp gpurun_out/${p}_hbm_traffic_pmc_$c.json profiles/${p}_hbm_traffic_pmc_$c.json; done
python3 - <<PY
import json
json.dump([json.load(open("profiles/${p}_hbm_traffic_pmc_%s.json" % c)) for c in ("c2", "c4", "c5")], open("profiles/hbm_traffic_pmc.json", "w"), indent=1)
PY
cp gpurun_out/${p}_parity.json profiles/parity_${r}.json
cp gpurun_out/${p}_parity_heavy_tailed.json profiles/parity_${r}_heavy_tailed.json
tail -n 3 gpurun_out/${b}_pytest.log > profiles/${b}_pytest_tail.txt
grep -h "^\[plain 1e-5\]\|^\[gradients\|^\[fp8 cross\|^\[lone crystal\|^\[basis stash\|^\[bench path\|^   \[fp64 ref\]\|^\[loss" gpurun_out/${b}_pytest.log > profiles/${b}_pytest_parity_lines.txt
cp gpurun_out/${b}_multistream.log profiles/${b}_multistream_optin.txt 2>/dev/null
