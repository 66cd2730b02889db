#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/gpurun_out/r03v; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/line.json 2> $out/trace.err < /dev/null
db=$(find $out/trace -name "*.db" | head -1)
[ -n "$db" ] && timeout 120 python $R/tools/prof_summary.py $db $out/r03v_bench_kernel_trace_stats.md "Round 3 (r03v): bench.py step" > /dev/null 2>&1 < /dev/null
rm -rf $out/trace
sed -n 5,6p $out/r03v_bench_kernel_trace_stats.md; grep -E "split_reduce|bn_finalize_bwd|gemm_reduce" $out/r03v_bench_kernel_trace_stats.md | cut -c1-110
