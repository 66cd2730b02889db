"""Per-wave instruction / wait account of a kernel family from the text tools/pmc_kernel.sh prints (dev tool).
usage: python tools/pmc_sq_table.py <pmc_kernel.sh output> <out.md>
Counters are per launch (mean over the recorded dispatches), summed over the chip.  SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count
quad-cycles (microarchitecture guide, gfx950 section), so their ratios are shares of a wave's resident time; GRBM_GUI_ACTIVE is summed over
the 8 XCDs and SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs."""
import collections
import re
import sys

src, out = sys.argv[1], sys.argv[2]
per = collections.OrderedDict()
cur = None
for line in open(src):
    m = re.match(r"^\s{3}(\w+)\s+(-?\d+)\s*$", line)
    if m and cur is not None:
        per[cur][m.group(1)] = float(m.group(2))
    elif line.strip() and not line.startswith(" "):
        cur = line.strip()
        per[cur] = {}
per = {k: v for k, v in per.items() if v.get("SQ_WAVES") and v.get("SQ_WAVE_CYCLES") and v.get("GRBM_GUI_ACTIVE")}


def share(v, a, b="SQ_WAVE_CYCLES"):
    return 100.0 * v.get(a, 0.0) / v[b] if v.get(b) else float("nan")


with open(out, "w") as f:
    f.write("| kernel | waves | wall cycles | VALU / wave | LDS / wave | VMEM rd / wave | SALU / wave | VALU active | LDS active | VMEM active | any active | "
            "issue wait | counter wait | LDS wait | waves / SIMD | MFMA busy | LDS bank conflict |\n|---|" + "---:|" * 16 + "\n")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
        w = v["SQ_WAVES"]
        wall = v["GRBM_GUI_ACTIVE"] / 8.0
        resident = 4.0 * v["SQ_WAVE_CYCLES"]                              # wave-cycles
        f.write("| `%s` | %d | %.0f | %.0f | %.0f | %.0f | %.0f | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.2f | %.1f %% | %.1f %% |\n" % (
            k[:70], w, wall, v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w, v.get("SQ_INSTS_VMEM_RD", 0) / w, v.get("SQ_INSTS_SALU", 0) / w,
            share(v, "SQ_ACTIVE_INST_VALU"), share(v, "SQ_ACTIVE_INST_LDS"), share(v, "SQ_INST_CYCLES_VMEM"), share(v, "SQ_ACTIVE_INST_ANY"),
            share(v, "SQ_WAIT_INST_ANY"), share(v, "SQ_WAIT_ANY"), share(v, "SQ_WAIT_INST_LDS"),
            resident / 1024.0 / wall if wall else float("nan"),
            100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / wall if wall else float("nan"),
            100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0.0))
print(open(out).read())
