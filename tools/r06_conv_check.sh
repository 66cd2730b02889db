# round 6: halo kernel re-pitch — parity, SQ counters of the convolution kernels in the step, plans re-measured, bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_conv.py tests/test_gpu_f16x2.py tests/test_gpu_nnkernels.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r06c/conv_tests.txt
python tools/make_pinned_plans.py > gpurun_out/r06c/make_plans.log 2>&1
cp sfmnext-impl_amd/plans/configB_resnet50_192x640_b12.json gpurun_out/r06c/
python bench.py --no-cpu-baseline > gpurun_out/r06c/bench_line.json 2> gpurun_out/r06c/bench.err
tools/pmc_kernel.sh gpurun_out/r06c/pmc "conv3x3_halo_kernel|conv_gemm_kernel" bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-diagnostics --no-roofline > gpurun_out/r06c/pmc_conv.txt 2>&1
python tools/pmc_sq_table.py gpurun_out/r06c/pmc_conv.txt gpurun_out/r06c/conv_sq_table.md
cat gpurun_out/r06c/conv_tests.txt; head -c 400 gpurun_out/r06c/bench_line.json; echo; grep halo gpurun_out/r06c/conv_sq_table.md | cut -c1-60,200-330
