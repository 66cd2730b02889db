# round 6: full GPU suite, the bench line, rocprofv3 kernel trace of the step, PMC traffic + SQ counters of the photometric kernels
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; tag=${1:-r06f}; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -6 > $out/gpu_suite.txt
python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.txt 2>&1
python bench.py > $out/bench_line.json 2> $out/bench.err
tools/profile_step.sh $out/prof $tag > $out/profile_step.log 2>&1
tools/pmc_photo.sh $out/pmc_photo --which fwd > $out/pmc_photo.log 2>&1
rm -rf $out/pmc_photo/*/
tail -3 $out/gpu_suite.txt; tail -2 $out/smoke.txt; head -c 300 $out/bench_line.json
