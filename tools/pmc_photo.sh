#!/bin/bash
# PMC passes over the photometric kernels (tools/bench_fused.py): one rocprofv3 run per counter group, kernel-trace only.
# usage: tools/pmc_photo.sh <outdir> [bench_fused args...]
set -u
out=$1; shift
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$out/$name -- python $R/tools/bench_fused.py --iters 30 ${ARGS} > $R/$out/$name.log 2>&1; }
ARGS="$*"
mkdir -p $R/$out
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run sq3 SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python $R/tools/pmc_table.py $R/$out photo_ > $R/$out/table.txt 2>&1
cat $R/$out/table.txt
