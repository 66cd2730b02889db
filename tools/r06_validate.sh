set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06b
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06b/gpu_suite.txt
python bench.py > gpurun_out/r06b/bench_line.json 2> gpurun_out/r06b/bench.err
tools/profile_step.sh gpurun_out/r06b/prof r06b > gpurun_out/r06b/profile_step.log 2>&1
tools/pmc_photo.sh gpurun_out/r06b/pmc_photo --which fwd > gpurun_out/r06b/pmc_photo.log 2>&1
rm -rf gpurun_out/r06b/pmc_photo/*/
tail -3 gpurun_out/r06b/gpu_suite.txt; head -c 1500 gpurun_out/r06b/bench_line.json
