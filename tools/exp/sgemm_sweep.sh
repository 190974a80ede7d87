#!/bin/bash
# usage (GPU box): tools/exp/sgemm_sweep.sh "<variants>" "<shape filters>" [mode mask] -- tools/exp/_bin/sgemm_bench_<variant> per shape filter
for f in $2; do for v in $1; do printf "%-8s " $v; timeout -k 10 120 tools/exp/_bin/sgemm_bench_$v "$f" ${3:-4} | cut -c1-200 || exit 1; done; done
