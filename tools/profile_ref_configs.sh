#!/bin/bash
# Driver-format bench lines + rocprofv3 kernel-trace summaries of the reference's own KITTI configurations (args_files/hisfog/kitti/*.txt:
# their network / loss flags verbatim — every one of them trains with --use_stereo --diff_lr, i.e. three source frames —; paths, weight
# folders and evaluation flags dropped, synthetic frames) on one MI355X.  Batch: BASELINE.json's where it names one, else the file's.
# usage: tools/profile_ref_configs.sh <outdir> <tag> [names...]      (copy what is to be judged into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/$1; tag=$2; shift 2; only="$*"
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() {   # name, SQD_BENCH_EXTRA, workload line
  if [ -n "$only" ] && ! echo " $only " | grep -q " $1 "; then return; fi
  export SQD_BENCH_EXTRA="$2" SQD_BENCH_WORKLOAD="$3"
  timeout 900 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-diagnostics > $out/${tag}_$1_bench_line.json 2> $out/$1.err
  timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace_$1 -- python $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-roofline --no-diagnostics > /dev/null 2> $out/$1.trace.err
  db=$(find $out/trace_$1 -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/prof_summary.py $db $out/${tag}_$1_kernel_trace_stats.md "Round ${tag:1:2} ($tag): $3 — rocprofv3 --kernel-trace --stats -- SQD_BENCH_EXTRA='$2' python bench.py --steps 12 --warmup 5" "bins_fwd_" "$3" > /dev/null
  unset SQD_BENCH_EXTRA SQD_BENCH_WORKLOAD
  rm -rf $out/trace_$1          # (the raw trace is ~30 MB per run; gpurun copies at most 64 MB back)
  head -c 500 $out/${tag}_$1_bench_line.json; echo; sed -n 3,8p $out/${tag}_$1_kernel_trace_stats.md | cut -c1-150
}
COMMON="--min_depth 0.001 --max_depth 80.0 --diff_lr --use_stereo"
run refB "--backbone resnet_lite --num_layers 50 --num_features 256 --model_dim 32 --patch_size 16 --dim_out 64 --query_nums 64 --height 192 --width 640 --batch_size 16 $COMMON" \
    "args_files/hisfog/kitti/resnet_192x640.txt (ResNet-50 + Lite_Depth_Decoder_QueryTr, 192x640, batch 16, frames 0 -1 1 s, --diff_lr), fp32, 1 x MI355X"
run refC "--backbone resnet_lite --num_layers 50 --num_features 256 --model_dim 32 --patch_size 20 --dim_out 128 --query_nums 128 --height 320 --width 1024 --batch_size 8 --min_depth 0.01 --max_depth 80.0 --diff_lr --use_stereo" \
    "configs[2] as args_files/hisfog/kitti/resnet_320x1024.txt has it (ResNet-50 + Lite_Depth_Decoder_QueryTr, 320x1024, frames 0 -1 1 s, --diff_lr; batch 8 per GPU as BASELINE.json says, the file's is 16), fp32, 1 x MI355X"
run refD "--backbone tf_efficientnet_b5_ap --height 320 --width 1024 --batch_size 16 --model_dim 32 --patch_size 32 --dim_out 128 --query_nums 128 --dec_channels 512 256 128 64 32 --sqd_bf16 $COMMON" \
    "configs[3] as args_files/hisfog/kitti/effb5_320x1024.txt has it (EfficientNet-b5, 320x1024, batch 16, patch 32 / Q 128 / dim_out 128, frames 0 -1 1 s, --diff_lr; --model_type dropped: not an options.py flag in the reference either), bf16 convolution operands, 1 x MI355X"
run refE "--backbone convnext_large --height 320 --width 1024 --batch_size 16 --model_dim 32 --patch_size 32 --dim_out 64 --query_nums 64 --dec_channels 1024 512 256 128 $COMMON" \
    "configs[4] trunk as args_files/hisfog/kitti/cvnXt_L_320x1024.txt has it (ConvNeXt-L U-Net in the self-supervised trainer, 320x1024, batch 16, frames 0 -1 1 s, --diff_lr), fp32, 1 x MI355X"
