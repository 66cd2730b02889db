#!/bin/bash
# SQ counters of the kernels of a small bench script (dev): tools/pmc_kernel.sh <outdir> <kernel-name-regex> <python script + args...>
# (separate --pmc passes with --kernel-trace only, as the microarchitecture guide prescribes)
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; pat=$2; shift 2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python "$@" > $out/$tag.log 2>&1 < /dev/null
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"($pat)\w*(<[^>]*>)?", k)
        if not m: continue
        k = m.group(0)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]): print("   %-36s %14.0f" % (c, agg[k][c] / cnt[k][c]))
PY
rm -rf $out/*/
