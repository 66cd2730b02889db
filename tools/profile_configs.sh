#!/bin/bash
# Driver-format bench lines + rocprofv3 kernel-trace summaries of BASELINE.json configs[2..4] on one MI355X.
# usage: tools/profile_configs.sh <outdir> <tag>      (copy what is to be judged into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; tag=$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() {   # name, SQD_BENCH_EXTRA, workload line
  export SQD_BENCH_EXTRA="$2"
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-diagnostics > $out/${tag}_$1_bench_line.json 2> $out/$1.err
  rocprofv3 --kernel-trace --stats -d $out/trace_$1 -- python $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-roofline --no-diagnostics > /dev/null 2> $out/$1.trace.err
  db=$(find $out/trace_$1 -name "*.db" | head -1)
  python $R/tools/prof_summary.py $db $out/${tag}_$1_kernel_trace_stats.md "Round ${tag:1:2} ($tag): $3 — rocprofv3 --kernel-trace --stats -- SQD_BENCH_EXTRA='$2' python bench.py --steps 12 --warmup 5" "photo_tile_kernel<1" "$3" > /dev/null
  unset SQD_BENCH_EXTRA
  rm -rf $out/trace_$1          # (the raw trace is ~30 MB per run; gpurun copies at most 64 MB back)
  head -c 700 $out/${tag}_$1_bench_line.json; echo; sed -n 3,14p $out/${tag}_$1_kernel_trace_stats.md | cut -c1-150
}
run configC "--backbone resnet_lite --num_layers 50 --height 320 --width 1024 --batch_size 8 --patch_size 20 --query_nums 128 --dim_out 128 --min_depth 0.01" "configs[2]: ResNet-50 + Lite_Depth_Decoder_QueryTr, 320x1024, batch 8, Q 128 / dim_out 128 / patch 20 (args_files/hisfog/kitti/resnet_320x1024.txt), fp32, 1 x MI355X"
run configD "--backbone eff_b5 --height 320 --width 1024 --batch_size 8 --sqd_bf16 --model_dim 32 --patch_size 20" "configs[3]: EfficientNet-b5, 320x1024, batch 8, bf16 convolution operands, 1 x MI355X"
run configE "--backbone convnext_large --height 320 --width 1024 --batch_size 4 --model_dim 32 --patch_size 32" "configs[4] trunk: ConvNeXt-L U-Net in the self-supervised trainer, 320x1024, batch 4, fp32, 1 x MI355X"
python $R/tools/bench_finetune.py --bs 4 --steps 20 --warmup 3 > $out/${tag}_configE_finetune_bench_line.json 2> $out/finetune.err; head -c 600 $out/${tag}_configE_finetune_bench_line.json
