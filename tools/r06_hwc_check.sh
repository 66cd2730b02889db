# round 6: channels_last source frames through the photometric kernels and the captured step — plans re-pinned (sqd_common.h lost the
# photometric declarations), the whole GPU suite, then a bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06h
timeout 900 python tools/make_pinned_plans.py gpurun_out/r06h/configB_plans.json > gpurun_out/r06h/plans.log 2>&1
cp gpurun_out/r06h/configB_plans.json sfmnext-impl_amd/plans/configB_resnet50_192x640_b12.json
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06h/tests.txt
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r06h/bench.json 2> gpurun_out/r06h/bench.err
tail -2 gpurun_out/r06h/plans.log | cut -c1-300; cat gpurun_out/r06h/tests.txt | cut -c1-250; cut -c1-400 gpurun_out/r06h/bench.json; tail -2 gpurun_out/r06h/bench.err | cut -c1-300
