#!/bin/bash
# A/B of two pinned plan files on one box
cd $GRAFT_REPO_ROOT
P=sfmnext-impl_amd/plans/configB_resnet50_192x640_b12.json
cp $P /tmp/plans_r1.json
SQD_TUNE_ROUNDS=3 python tools/make_pinned_plans.py /tmp/plans_r3.json 2>&1 | tail -1
run() { cp $1 $P; python bench.py --no-cpu-baseline --no-roofline --no-diagnostics --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['conv_arith']['plans'][:12])"; }
for i in 1 2 3; do run /tmp/plans_r1.json; run /tmp/plans_r3.json; done
cp /tmp/plans_r3.json gpurun_out/pinned_plans_r3.json
cp /tmp/plans_r1.json $P
