"""Table of the depth-head kernels' HBM-side traffic and durations from the rocprofv3 output tools/pmc_heads.sh collected.
usage: python tools/pmc_heads.py <outdir> "<B Q D h w E>"  -> markdown on stdout
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; corrected with the calibration kernels of tools/pmc_calib.py (dword-wide accesses, as the
microarchitecture guide's HBM section prescribes: the counters are uncalibrated for other widths)."""
import collections
import csv
import glob
import sys

root, shape = sys.argv[1], sys.argv[2]
B, Q, D, h, w, E = (int(v) for v in shape.split())
N = h * w
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)


def avg(table, sub, key=None):
    for k, v in table.items():
        if sub in k:
            vals = v[key] if key else v
            vals = vals[len(vals) // 2:]
            return sum(vals) / len(vals), k
    return None, None


GIB = float(1 << 30)
cr, _ = avg(acc, "calib_read_dword", "FETCH_SIZE")
cw, _ = avg(acc, "calib_write_dword", "WRITE_SIZE")
fr, fw = GIB / (cr * 1024.0), GIB / (cw * 1024.0)
px = B * N
# algorithmic bytes per launch: y / g_y are [B,Q,N] planes, x [B,N,E]
ALG = {"sql_fwd32_kernel": ("Self Query Layer forward: read x 4E, write y 4Q per pixel", px * 4 * (E + Q)),
       "sql_bwd32q_kernel": ("Self Query Layer backward: read x 4E, g_y 4Q (+ y recomputed), write g_x 4E", px * 4 * (2 * E + Q)),
       "sql_bwd32_kernel": ("Self Query Layer backward: read x 4E, g_y 4Q (+ y recomputed), write g_x 4E", px * 4 * (2 * E + Q)),
       "bins_fwd_kernel": ("bins head forward: read the energy maps 4Q, write pred 4 per pixel", px * 4 * (Q + 1)),
       "bins_bwd_kernel": ("bins head backward: read the energy maps 4Q, g 4, write dE 4Q per pixel", px * 4 * (2 * Q + 1)),
       "bins_fwd_h_kernel": ("bins head forward on two-term fp16 operands (round 5)", px * 4 * (Q + 1)),
       "bins_bwd_h_kernel": ("bins head backward, logits and dE on two-term fp16 operands (round 5)", px * 4 * (2 * Q + 1))}
FLOP = {"bins_fwd_kernel": 2.0 * px * Q * D, "bins_bwd_kernel": 6.0 * px * Q * D, "bins_fwd_h_kernel": 2.0 * px * Q * D, "bins_bwd_h_kernel": 6.0 * px * Q * D, "sql_fwd32_kernel": 4.0 * px * Q * E,
        "sql_bwd32q_kernel": 8.0 * px * Q * E, "sql_bwd32_kernel": 8.0 * px * Q * E}
print("# Depth-head kernels at B=%d Q=%d D=%d E=%d on %dx%d maps: PMC traffic and durations (one MI355X)\n" % (B, Q, D, E, h, w))
print("`tools/pmc_heads.sh`: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/bench_bins.py / tools/bench_sql.py, corrected by the "
      "dword-width calibration kernels (factors %.3f / %.3f); durations from a --kernel-trace pass without counters.  fp32 MFMA floor = the kernel's "
      "matrix flops at 157.3 TFLOP/s; HBM floor = algorithmic bytes at 8 TB/s.\n" % (fr, fw))
print("| kernel | us | algorithmic MB | PMC traffic MB (x algorithmic) | achieved TB/s (of 8) | HBM floor us | fp32 MFMA floor us | matrix TFLOP/s |")
print("|---|---:|---:|---:|---:|---:|---:|---:|")
for sub, (what, alg) in ALG.items():
    t, name = avg(dur, sub)
    f, _ = avg(acc, sub, "FETCH_SIZE")
    wv, _ = avg(acc, sub, "WRITE_SIZE")
    if t is None or f is None:
        continue
    traffic = f * 1024.0 * fr + wv * 1024.0 * fw
    print("| `%s` (%s) | %.1f | %.1f | %.1f (%.2f) | %.2f (%.3f) | %.1f | %.1f | %.1f |"
          % (name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0], what, t, alg / 1e6, traffic / 1e6, traffic / alg, alg / t / 1e6, alg / t / 1e6 / 8.0,
             alg / 8e6, FLOP[sub] / 157.3e6, FLOP[sub] / t / 1e6))
