"""HBM traffic of one training step from two rocprofv3 passes over bench.py (--pmc FETCH_SIZE, --pmc WRITE_SIZE; counters only + kernel trace).
usage: python tools/pmc_step_traffic.py <dir> <out.md> <ms per step of the unprofiled bench line>
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; reads are scaled by the calibration factor of profiles/r05m_pmc_traffic.json (2.0 on gfx950 for
this counter, 1.0 for writes: tools/pmc_calib.py against kernels of known traffic, as MI355X_MICROARCH.md's HBM section prescribes).  A step =
the dispatches between two adam_kernel launches; the recorded graph replays (the steps after the three eager ones) are averaged."""
import collections
import csv
import glob
import json
import os
import sys

root, out, ms = sys.argv[1], sys.argv[2], float(sys.argv[3])
cal = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05m_pmc_traffic.json")))["calibration_factor"]
fac = {"FETCH_SIZE": cal["read_f4"], "WRITE_SIZE": cal["write_f4"]}
per_step = {}
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "adam_kernel" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    keep = steps[3:]                                     # replayed steps (the run: 3 eager steps, capture, replays; the collection records the first few replays only)
    assert len(keep) >= 1 and max(len(s) for s in keep) - min(len(s) for s in keep) < 30, [len(s) for s in steps]
    tot = 0.0
    for s in keep:
        for r in s:
            b = float(r["Counter_Value"]) * 1024.0 * fac[counter] / len(keep)
            tot += b
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            fam[k][counter] += b
    per_step[counter] = (tot, min(len(s) for s in keep))
rd, wr = per_step["FETCH_SIZE"][0], per_step["WRITE_SIZE"][0]
with open(out, "w") as f:
    f.write("| | per step |\n|---|---:|\n| kernel launches | %d |\n| HBM read (FETCH_SIZE x %.1f) | %.2f GB |\n| HBM written (WRITE_SIZE x %.1f) | %.2f GB |\n"
            "| total | %.2f GB |\n| over the unprofiled step (%.3f ms) | %.2f TB/s = %.2f of the 8 TB/s peak |\n\n" % (
                per_step["FETCH_SIZE"][1], fac["FETCH_SIZE"], rd / 1e9, fac["WRITE_SIZE"], wr / 1e9, (rd + wr) / 1e9, ms, (rd + wr) / ms / 1e9, (rd + wr) / ms / 1e9 / 8.0))
    f.write("| kernel family | read GB | written GB | share of the step's traffic |\n|---|---:|---:|---:|\n")
    for k, v in sorted(fam.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:24]:
        f.write("| `%s` | %.2f | %.2f | %.1f %% |\n" % (k[:60], v["FETCH_SIZE"] / 1e9, v["WRITE_SIZE"] / 1e9, 100 * (v["FETCH_SIZE"] + v["WRITE_SIZE"]) / (rd + wr)))
print(open(out).read())
