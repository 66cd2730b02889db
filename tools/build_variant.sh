#!/bin/bash
# Build a variant of libsqd.so that differs from the tree's build in ONE kernel file compiled with extra flags, for same-box A/B runs:
#   tools/build_variant.sh <name> <file.hip> [-DFLAG ...]  ->  tools/bin/libsqd_<name>.so  (python tools/bench_fused.py --lib ...)
set -e
name=$1; file=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); B=$R/sfmnext-impl_amd/csrc/_build
mkdir -p $R/tools/bin
base=$(basename $file .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c $R/sfmnext-impl_amd/csrc/$base.hip -o /tmp/variant_${name}_$base.o
objs=$(ls $B/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_${name}_$base.o -o $R/tools/bin/libsqd_$name.so
echo built $R/tools/bin/libsqd_$name.so
