#!/bin/bash
# Same-box A/B of the training step between the tree's libsqd.so and a variant (tools/build_variant.sh): alternating fresh processes.
#   tools/ab_lib_bench.sh "<variant name> [<variant name> ...]" [rounds]
R=$(cd $(dirname $0)/.. && pwd); v=$1; n=${2:-3}
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
print(sys.argv[1] if False else "", d["ms_per_step"], d["value"])
PY
}
cd $R
for i in $(seq $n); do echo -n "tree: "; run tree; for w in $v; do echo -n "$w: "; run tools/bin/libsqd_$w.so; done; done
