#!/usr/bin/env python3
"""Same-box A/B of bench.py with a product module attribute flipped (dev tool): alternates the two settings in fresh processes.
usage: python tools/ab_bench.py nnkernels.FUSE_BN_BWD_STATS=False | "nnkernels.TUNE_SPACE[wgrad_rows]=False" [--rounds 3] [bench.py arguments ...]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "sfmnext-impl_amd"))
setting = %(setting)r
if setting:
    mod, _, rest = setting.partition(".")
    name, _, val = rest.partition("=")
    import importlib
    m = importlib.import_module("sqd." + mod)
    if "[" in name:                                   # nnkernels.TUNE_SPACE[wgrad_rows]=False
        name, _, key = name.partition("[")
        getattr(m, name)[key.rstrip("]")] = eval(val)
    else:
        setattr(m, name, eval(val))
import bench
sys.argv = ["bench.py"] + %(args)r
bench.main()
"""


def run(setting, args):
    r = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO, "setting": setting, "args": args}], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise SystemExit(r.stderr[-3000:])
    return json.loads(line[-1])["ms_per_step"]


def main():
    setting = sys.argv[1]
    rest = sys.argv[2:]
    rounds = 3
    if rest[:1] == ["--rounds"]:
        rounds, rest = int(rest[1]), rest[2:]
    args = rest or ["--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-roofline"]
    a, b = [], []
    for _ in range(rounds):
        a.append(run("", args))
        b.append(run(setting, args))
    print("default            ms/step:", a)
    print("%-18s ms/step:" % setting, b)
    print("mean difference (setting - default): %+.3f ms" % (sum(b) / len(b) - sum(a) / len(a)))


if __name__ == "__main__":
    main()
