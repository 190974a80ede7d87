#!/bin/bash
# usage (here): tools/exp/build_exp.sh <source.hip> <tag> "<extra hipcc flags>"
# Builds tools/exp/ab/lib_<tag>.so = the in-tree library with ONE source recompiled with extra flags (A/B and timing experiments).
# The "remove a part" hooks (-DARREAU_EXP=<bits>: builds that compute wrong numbers on purpose and never ship) are NOT in the
# product sources: when the flags mention ARREAU_EXP, tools/exp/arreau_exp_hooks.patch is applied to a scratch copy of
# csrc/ and the source is compiled from there.
set -e
src=$1; tag=$2; extra=$3
cd "$(dirname "$0")/../.."
root=$(pwd)
csrc=arreau_amd/csrc
mkdir -p tools/exp/ab /tmp/arreau_exp_$tag
python -m arreau_amd.build >/dev/null 2>&1
from=$csrc
case "$extra" in *ARREAU_EXP*)
  rm -rf /tmp/arreau_exp_$tag/tree && mkdir -p /tmp/arreau_exp_$tag/tree/arreau_amd && cp -r $csrc /tmp/arreau_exp_$tag/tree/arreau_amd/ && cp -r include /tmp/arreau_exp_$tag/tree/
  (cd /tmp/arreau_exp_$tag/tree && patch -p1 -s < "$root/tools/exp/arreau_exp_hooks.patch")
  from=/tmp/arreau_exp_$tag/tree/arreau_amd/csrc;;
esac
slp=""; case $src in edge_f16.hip|node_f16.hip|node_f16m.hip) slp="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-inline-asm -Wno-unused-but-set-variable -Wno-misleading-indentation -ffp-contract=on $slp $extra \
    -c $from/$src -o /tmp/arreau_exp_$tag/${src%.hip}.o
objs=""
for o in $csrc/*.o; do
  if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs /tmp/arreau_exp_$tag/${src%.hip}.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/ab/lib_$tag.so $objs
echo built tools/exp/ab/lib_$tag.so
