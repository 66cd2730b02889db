#!/usr/bin/env python3
"""Sum a prof_summary.py kernel table by kernel family (dev tool).  usage: python tools/trace_families.py a.md [b.md]"""
import re
import sys

FAMS = [("conv fwd", r"conv_gemm_kernel<0|conv3x3_halo_kernel<0"), ("conv dgrad", r"conv_gemm_kernel<1|conv3x3_halo_kernel<1"),
        ("conv wgrad", r"conv_wgrad"), ("split / gemm reduce", r"split_reduce|gemm_reduce"), ("batchnorm", r"bn_"),
        ("photometric", r"photo_|smooth|depth_up|pose_mats"), ("heads (sql, bins)", r"sql_|bins_|bin_centers"),
        ("transformer", r"mha_|ffn_|addln|colsum|ln_"), ("adam", r"adam"), ("upcat / pool / act / amax", r"upcat|maxpool|act_bwd|amax|space_to_depth")]


def load(path):
    tot, launches = {}, {}
    for line in open(path):
        m = re.match(r"\|\s*([\d.]+)\s*\|\s*([\d.]+)\s*\|\s*([\d.]+)\s*\|\s*`(.*)`", line)
        if not m:
            continue
        us, calls, name = float(m.group(1)), float(m.group(2)), m.group(4)
        fam = next((f for f, pat in FAMS if re.search(pat, name)), "other")
        tot[fam] = tot.get(fam, 0.0) + us
        launches[fam] = launches.get(fam, 0.0) + calls
    return tot, launches


cols = [load(p) for p in sys.argv[1:]]
print("%-28s" % "family" + "".join("%22s" % p.split("/")[-1][:20] for p in sys.argv[1:]))
for fam in [f for f, _ in FAMS] + ["other"]:
    print("%-28s" % fam + "".join("%14.0f us %4.0f" % (t.get(fam, 0), l.get(fam, 0)) for t, l in cols))
print("%-28s" % "total" + "".join("%14.0f us %4.0f" % (sum(t.values()), sum(l.values())) for t, l in cols))
