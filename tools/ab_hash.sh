#!/bin/bash
# usage (GPU box): tools/ab_hash.sh -- output hashes of one score evaluation with the previous build (tools/exp/ab/lib_prev.so) and the in-tree one
cd $GRAFT_REPO_ROOT
for cfg in "256 20" "3 7"; do
  ARREAU_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/${AB_LIB:-lib_prev.so} timeout -k 10 200 python tools/determinism.py $cfg | tail -n 1 | sed "s/^/prev $cfg: /"
  timeout -k 10 200 python tools/determinism.py $cfg | tail -n 1 | sed "s/^/cur  $cfg: /"
done
