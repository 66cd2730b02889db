# round 6: packed fp16 coefficient records — parity of the photometric gradients, timing, traffic; SQ counters of the convolution kernels
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_photometric.py tests/test_gpu_golden_replay.py tests/test_gpu_stereo.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r06d/photo_tests.txt
python tools/bench_fused.py --which fwd,coef,bwd > gpurun_out/r06d/bench_fused.txt 2>&1
tools/pmc_kernel.sh gpurun_out/r06d/pmc "conv3x3_halo_kernel|conv_gemm_kernel" $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-diagnostics --no-roofline > gpurun_out/r06d/pmc_conv.txt 2>&1
python tools/pmc_sq_table.py gpurun_out/r06d/pmc_conv.txt gpurun_out/r06d/conv_sq_table.md
cat gpurun_out/r06d/photo_tests.txt; tail -2 gpurun_out/r06d/bench_fused.txt; grep halo gpurun_out/r06d/conv_sq_table.md | awk -F'|' '{print $2, $4, $(NF-1)}'
