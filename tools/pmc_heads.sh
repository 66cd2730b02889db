#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes, with the access-width calibration kernels) and kernel durations of the depth
# head's kernels — Self Query Layer and bins head, forward and backward — at the shape given: tools/pmc_heads.sh <outdir> [B Q D h w E]
# (configs[2]: 8 128 128 160 512 32)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
B=${1:-8}; Q=${2:-128}; D=${3:-128}; h=${4:-160}; w=${5:-512}; E=${6:-32}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc/calib_$c -- python $R/tools/pmc_calib.py > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc/bins_$c -- python $R/tools/bench_bins.py $B $Q $D $h $w > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc/sql_$c -- python $R/tools/bench_sql.py $B $E $Q $h $w > /dev/null 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $out/trace/bins -- python $R/tools/bench_bins.py $B $Q $D $h $w > $out/bins.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/trace/sql -- python $R/tools/bench_sql.py $B $E $Q $h $w > $out/sql.log 2>&1
python $R/tools/pmc_heads.py $out "$B $Q $D $h $w $E" > $out/heads_pmc.md 2> $out/heads_pmc.err
rm -rf $out/pmc $out/trace
cat $out/heads_pmc.md
