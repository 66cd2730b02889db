"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES` pass over bench.py.
usage: python tools/pmc_mfma_table.py <dir> <out.md> [first-dispatch-fraction-to-skip=0.5]
GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles / 8 = wall cycles of the dispatch); SQ_VALU_MFMA_BUSY_CYCLES is summed over
the 1024 SIMDs, so utilisation = MFMA_BUSY / 1024 / (GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import sys

root, out = sys.argv[1], sys.argv[2]
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = []
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
cut = ids[int(len(ids) * skip)]                      # steady state: skip warm-up / tuning dispatches
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    d = int(r["Dispatch_Id"])
    if d < cut:
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (d, k) not in seen:
        seen.add((d, k))
        cnt[k] += 1
tot_busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in per.values())
tot_act = sum(v["GRBM_GUI_ACTIVE"] for v in per.values())
with open(out, "w") as f:
    f.write("| kernel | dispatches | GPU-active share | MFMA busy / active |\n|---|---:|---:|---:|\n")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:25]:
        act = v["GRBM_GUI_ACTIVE"] / 8.0
        util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / act if act else 0.0
        f.write("| `%s` | %d | %.1f %% | %.1f %% |\n" % (k, cnt[k], 100 * v["GRBM_GUI_ACTIVE"] / tot_act, 100 * util))
    f.write("\nall kernels of the window: MFMA busy %.1f %% of the GPU-active cycles\n" % (100 * tot_busy / 1024.0 / (tot_act / 8.0)))
print(open(out).read())
