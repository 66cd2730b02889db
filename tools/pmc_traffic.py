"""Corrected HBM traffic of the fused warp+SSIM kernel from rocprofv3 PMC passes.
usage: python tools/pmc_traffic.py <dir with the --pmc FETCH_SIZE / WRITE_SIZE output of pmc_calib.py and bench_fused.py> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Correction = known bytes / reported bytes of the calibration
kernel with the same access width (dword loads / dword stores), as MI355X_MICROARCH.md's HBM section prescribes."""
import collections
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))


matched = {}


def avg(sub, counter):
    for k, cs in acc.items():
        if sub in k and counter in cs:
            v = cs[counter][len(cs[counter]) // 2:]          # second half of the dispatches: caches and clocks settled
            matched[sub] = k[:k.index(">(") + 1] if ">(" in k else k[:100]
            return sum(v) / len(v) * 1024.0
    raise SystemExit("no %s for %s under %s" % (counter, sub, root))


GIB = float(1 << 30)
cal = {"read_dword": GIB / avg("calib_read_dword", "FETCH_SIZE"), "read_f4": GIB / avg("calib_read_f4", "FETCH_SIZE"),
       "write_dword": GIB / avg("calib_write_dword", "WRITE_SIZE"), "write_f4": GIB / avg("calib_write_f4", "WRITE_SIZE")}
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash: the record is valid for this revision of the kernel sources only)
res = {"kernel_source_hash": bench.kernel_source_hash(), "judged_kernel": "photo_tile_kernel<1>",
       "calibration_factor": {k: round(v, 4) for k, v in cal.items()}, "kernels": {}}
for name in ("photo_tile_kernel<1>", "photo_tile_kernel<0>", "photo_tile_kernel<2>", "photo_bwd_tile_kernel"):
    try:
        sub = name.rstrip(">")                  # (the kernels carry a second template argument — waves per workgroup — in their names)
        fr, wr = avg(sub, "FETCH_SIZE"), avg(sub, "WRITE_SIZE")
    except SystemExit:
        continue
    res["kernels"][name] = {"instantiation": matched.get(sub), "fetch_reported_bytes": round(fr), "write_reported_bytes": round(wr),
                            "fetch_corrected_bytes": round(fr * cal["read_dword"]), "write_corrected_bytes": round(wr * cal["write_dword"]),
                            "traffic_bytes": round(fr * cal["read_dword"] + wr * cal["write_dword"]),
                            "note": "dword-width calibration for both directions (the kernels' gathers are 8- and 16-byte, their row loads and stores 4- to 16-byte)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
