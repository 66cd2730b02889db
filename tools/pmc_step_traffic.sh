#!/bin/bash
# HBM traffic of the training step (dev): tools/pmc_step_traffic.sh <outdir>  -> <outdir>/step_traffic.md  (two counter passes + one plain bench line)
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-diagnostics > $out/bench_line.json 2> $out/bench.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-diagnostics > $out/$c.log 2>&1 < /dev/null
done
ms=$(python -c "import json,sys; print(json.loads([l for l in open('$out/bench_line.json') if l.startswith('{')][-1])['ms_per_step'])")
python $R/tools/pmc_step_traffic.py $out $out/step_traffic.md $ms
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
