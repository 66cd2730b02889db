#!/bin/bash
# kernel durations of the Self Query Layer / bins head benches under rocprofv3 (dev): tools/prof_sql.sh [bench_sql.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_sql
timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_sql -- python $R/tools/bench_sql.py "$@" > /tmp/prof_sql.log 2>&1 < /dev/null
db=$(find /tmp/prof_sql -name "*.db" | head -1)
tail -2 /tmp/prof_sql.log
[ -n "$db" ] && timeout 60 python $R/tools/kernel_durations.py "$db" sql < /dev/null
