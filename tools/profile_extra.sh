#!/bin/bash
# kernel-trace summary of bench.py with SQD_BENCH_EXTRA (other configurations): tools/profile_extra.sh <outdir> <tag> "<extra args>"
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; tag=$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
SQD_BENCH_EXTRA="$3" rocprofv3 --kernel-trace --stats -d $out/trace_$tag -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline > $out/${tag}_bench_line.json 2> $out/trace_$tag.err
db=$(find $out/trace_$tag -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $out/${tag}_kernel_trace_stats.md "Round 2 ($tag): bench.py with SQD_BENCH_EXTRA='$3'" > /dev/null
head -45 $out/${tag}_kernel_trace_stats.md | cut -c1-170
