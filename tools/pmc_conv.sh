#!/bin/bash
# PMC counters of the convolution kernels over tools/abl_conv.py (dev): tools/pmc_conv.sh <outdir> <lib>
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; lib=$R/$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $out/avail.txt 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python $R/tools/abl_conv.py $lib > $out/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections, os
out = "$out"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_gemm" not in k: continue
        k = k[k.index("conv_gemm_kernel"):][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]): print("   %-36s %14.0f  (per launch, %d launches)" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
PY
