#!/bin/bash
# kernel durations of the bins head bench under rocprofv3 (dev): tools/prof_bins.sh [B Q D h w]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_bins
timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_bins -- python $R/tools/bench_bins.py "$@" > /tmp/prof_bins.log 2>&1 < /dev/null
db=$(find /tmp/prof_bins -name "*.db" | head -1)
grep fwd /tmp/prof_bins.log | tail -1
[ -n "$db" ] && timeout 60 python $R/tools/kernel_durations.py "$db" bins < /dev/null
