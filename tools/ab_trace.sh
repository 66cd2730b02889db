#!/bin/bash
# Same-box kernel-trace A/B of bench.py under two SQD_BENCH_EXTRA settings (live plans both): family sums side by side.
# usage: tools/ab_trace.sh <outdir> <tag> "<extra A>" "<extra B>"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; tag=$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for v in A B; do
  [ $v = A ] && extra="$3" || extra="$4"
  SQD_BENCH_LIVE_PLANS=1 SQD_BENCH_EXTRA="$extra" rocprofv3 --kernel-trace --stats -d $out/trace_$v -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-diagnostics --no-roofline > $out/${tag}_$v.json 2> $out/$v.err
  db=$(find $out/trace_$v -name "*.db" | head -1)
  python $R/tools/prof_summary.py $db $out/${tag}_${v}_kernel_trace_stats.md "$tag $v: bench.py with '$extra'" > /dev/null
  rm -rf $out/trace_$v
done
python $R/tools/trace_families.py $out/${tag}_A_kernel_trace_stats.md $out/${tag}_B_kernel_trace_stats.md
