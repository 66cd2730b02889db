#!/bin/bash
# Build libsqd.so from another revision's kernel sources next to the current one, for same-box A/B runs:
#   tools/build_alt_lib.sh <git-rev> -> tools/bin/libsqd_<rev>.so   (use: python tools/bench_fused.py --lib tools/bin/libsqd_<rev>.so)
set -e
rev=$1; R=$(cd $(dirname $0)/.. && pwd); tmp=$(mktemp -d)
mkdir -p $tmp/sfmnext-impl_amd $tmp/include $R/tools/bin
git -C $R archive $rev sfmnext-impl_amd/csrc include | tar -x -C $tmp
cd $tmp/sfmnext-impl_amd/csrc
ls *.hip | xargs -P 8 -I{} hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c {} -o {}.o
hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $R/tools/bin/libsqd_$rev.so
rm -rf $tmp; echo built $R/tools/bin/libsqd_$rev.so
