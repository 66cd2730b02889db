#!/usr/bin/env python3
"""Soak of the default multi-rank mode: N fresh processes, each builds a Trainer on a 1-rank RCCL communicator, trains 3 eager steps,
captures forward + backward + in-graph all-reduces + Adam into one hipGraph and replays it (tests/test_gpu_zz_multirank.py's
graph-overlap leg).  Prints one line per attempt and a summary; exit code 1 if any attempt failed.

    python tools/ddp_capture_soak.py [N=20]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests"), REPO, os.path.join(REPO, "sfmnext-impl_amd"), os.path.join(REPO, "tests", "golden")]
import test_gpu_zz_multirank as M  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    plain = M._run({}, ["--sqd_no_graph"], "plain")
    bad = 0
    for i in range(n):
        t0 = time.time()
        try:
            res = M._run(M._dist_env(), [], "attempt %d" % i)
            M._check_dist(res, graph=True)
            M._same_training(plain, res)
            print("attempt %2d ok   %.1f s  losses %s" % (i, time.time() - t0, ["%.5f" % v for v in res["losses"][-2:]]), flush=True)
        except AssertionError as e:
            bad += 1
            print("attempt %2d FAILED %.1f s\n%s" % (i, time.time() - t0, str(e)[-6000:]), flush=True)
    print("soak: %d / %d fresh in-graph RCCL captures trained like the single-process run" % (n - bad, n))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
