#!/bin/bash
# SQ / TCP counters of the photometric tile kernels (dev): tools/pmc_photo2.sh <outdir> [bench_fused.py arguments...]
# (separate --pmc passes with --kernel-trace only, as the microarchitecture guide prescribes)
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; shift
ARGS=${@:---iters 20 --which fwd,ident}
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python $R/tools/bench_fused.py $ARGS > $out/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"photo_\w+kernel(<[^>]*>)?", k)
        if not m: continue
        k = m.group(0)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]): print("   %-36s %14.0f" % (c, agg[k][c] / cnt[k][c]))
PY
