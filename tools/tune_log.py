#!/usr/bin/env python3
"""Plan timing of the first training step with the per-arithmetic bests printed (dev tool): which plan family wins per geometry and by how much.
usage: python tools/tune_log.py [bench.py arguments ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
from sqd import nnkernels  # noqa: E402

nnkernels.TUNE_SPACE["log"] = True
os.environ["SQD_BENCH_LIVE_PLANS"] = "1"
import bench  # noqa: E402

sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-diagnostics"])
bench.main()
