#!/bin/bash
# Same-box A/B of (library build, pinned plan file) pairs on the training step: alternating fresh processes.
#   tools/ab_plans_lib.sh <rounds> <lib|tree>:<plans.json> [<lib|tree>:<plans.json> ...]
R=$(cd $(dirname $0)/.. && pwd); n=$1; shift
P=$R/sfmnext-impl_amd/plans/configB_resnet50_192x640_b12.json
cp $P /tmp/plans_keep.json
run() { python - "$1" <<'PY'
import json, os, runpy, sys, io, contextlib
R = os.getcwd()
sys.path.insert(0, os.path.join(R, "sfmnext-impl_amd"))
from sqd import lib as _l
if sys.argv[1] != "tree":
    _l.SO_PATH = os.path.abspath(sys.argv[1]); _l.needs_build = lambda: False
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--no-diagnostics", "--steps", "80"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
    runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["config"]["conv_arith"].get("plans", "")[:40])
PY
}
cd $R
for i in $(seq $n); do
  for pair in "$@"; do
    lib=${pair%%:*}; plans=${pair#*:}
    cp $plans $P
    echo -n "$pair: "; run $lib
  done
done
cp /tmp/plans_keep.json $P
