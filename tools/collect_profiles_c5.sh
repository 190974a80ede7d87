#!/bin/bash
# usage (here, after `tools/gpu_final_c5.sh <tag>` ran on the GPU box): tools/collect_profiles_c5.sh <tag>
t=$1
for c in c5 c5_hidden200 c5_exact_gemm c5_one_kernel_per_operation; do cp gpurun_out/${t}_bench_$c.json profiles/${t}_bench_$c.json; done
cp gpurun_out/${t}_c5_kernel_stats.csv profiles/${t}_kernel_stats_bench_c5.csv
cp gpurun_out/${t}_c5_step.txt profiles/${t}_c5_step_launch_list_forward_backward.txt
cp gpurun_out/${t}_c5_step_full.txt profiles/${t}_c5_step_launch_list_with_optimizer.txt
cp gpurun_out/${t}_hbm_traffic_pmc_c5.json profiles/${t}_hbm_traffic_pmc_c5.json
tail -n 3 gpurun_out/${t}_pytest_training.log > profiles/${t}_pytest_training_tail.txt
python3 - <<PY
import json
old = json.load(open("profiles/hbm_traffic_pmc.json"))
new = json.load(open("profiles/${t}_hbm_traffic_pmc_c5.json"))
json.dump([e for e in old if e.get("config") != "c5"] + [new], open("profiles/hbm_traffic_pmc.json", "w"), indent=1)
PY
