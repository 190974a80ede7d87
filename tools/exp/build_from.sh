#!/bin/bash
# usage (here): tools/exp/build_from.sh <path/to/modified/copy/of/source.hip> <tag> ["<extra hipcc flags>"]
# Links tools/exp/ab/lib_<tag>.so = the in-tree library with ONE object replaced by the compilation of a MODIFIED COPY of its
# source (timing-only experiments: "what would the kernel cost without X" -- such copies compute wrong numbers on purpose and
# live outside the product tree, e.g. under /tmp).  Headers come from arreau_amd/csrc.
set -e
file=$1; tag=$2; extra=$3
cd "$(dirname "$0")/../.."
csrc=arreau_amd/csrc
src=$(basename $file)
mkdir -p tools/exp/ab /tmp/arreau_exp_$tag
python -m arreau_amd.build >/dev/null 2>&1
slp=""; case $src in edge_f16.hip|node_f16.hip|node_f16m.hip) slp="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-inline-asm -Wno-unused-but-set-variable -Wno-misleading-indentation -ffp-contract=on $slp $extra \
    -I $csrc -c $file -o /tmp/arreau_exp_$tag/${src%.hip}.o
objs=""
for o in $csrc/*.o; do
  if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs /tmp/arreau_exp_$tag/${src%.hip}.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/ab/lib_$tag.so $objs
echo built tools/exp/ab/lib_$tag.so
