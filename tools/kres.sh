#!/bin/bash
# tools/kres.sh <file.hip> [kernel-name-substring]: register / scratch / spill summary of every kernel of a source file (cross-compile, no GPU)
src=$1; pat=${2:-.}
out=/tmp/isa/$(basename ${src%.hip}).s
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I/root/repo/include -S --cuda-device-only $SQD_HIPCC_EXTRA $src -o $out 2>&1 | grep -v "argument unused"
awk '$1==".name:"{n=$2} $1==".private_segment_fixed_size:"{pv=$2} $1==".sgpr_count:"{sg=$2} $1==".sgpr_spill_count:"{ss=$2} $1==".vgpr_count:"{vg=$2} $1==".vgpr_spill_count:"{printf "%-100s vgpr %3d sgpr %3d scratch %4d vspill %3d sspill %3d\n", substr(n,1,100), vg, sg, pv, $2, ss}' $out | grep -i -- "$pat"
