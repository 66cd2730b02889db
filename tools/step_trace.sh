#!/bin/bash
# kernel-trace summary of the training step at HEAD, nothing else (dev): tools/step_trace.sh <outdir> <tag> [grep pattern]
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; tag=$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/trace_$tag -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/${tag}_bench_line.json 2> /tmp/trace_$tag.err
db=$(find /tmp/trace_$tag -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $out/${tag}_bench_kernel_trace_stats.md "Round 3 ($tag): bench.py step (rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline)" > /dev/null
rm -rf /tmp/trace_$tag
sed -n 5p $out/${tag}_bench_kernel_trace_stats.md
grep -E "${3:-bn_}" $out/${tag}_bench_kernel_trace_stats.md | cut -c1-120
