#!/bin/bash
# One-stop profile of HEAD on the GPU box: kernel trace of the training step (bench.py), PMC traffic passes of the photometric
# kernels with their calibration, summaries under <outdir> (copy what is to be judged into profiles/).
# usage: tools/profile_step.sh <outdir> <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; tag=$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-diagnostics > $out/${tag}_bench_line.json 2> $out/trace.err
db=$(find $out/trace -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $out/${tag}_bench_kernel_trace_stats.md "Round ${tag:1:2} ($tag): bench.py step (rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-diagnostics)" > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc/calib_$c -- python $R/tools/pmc_calib.py > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc/fused_$c -- python $R/tools/bench_fused.py --iters 30 --which fwd,ident,coef,bwd > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $out/pmc $out/${tag}_pmc_traffic.json > $out/pmc_traffic.log 2>&1
rm -rf $out/trace $out/pmc          # (raw traces: tens of MB; gpurun copies at most 64 MB back)
tail -c 1500 $out/${tag}_pmc_traffic.json
head -60 $out/${tag}_bench_kernel_trace_stats.md
