#!/bin/bash
# Copies the summaries tools/collect_profiles.sh left under gpurun_out/profiles/ (scratch) into profiles/ (tracked) under this
# round's prefix:  tools/publish_profiles.sh r02
set -euo pipefail
cd "$(dirname "$0")/.."
r=${1:?round prefix, e.g. r02}
src=gpurun_out/profiles
declare -A map=( [bench.json]=bench.json [bench_prof.json]=bench_profiled_run.json [bench_kernel_stats.txt]=bench_kernel_stats.txt
  [bench_window_phases.txt]=bench_window_phases.txt [pmc_fetch_size.txt]=pmc_fetch_size.txt [pmc_write_size.txt]=pmc_write_size.txt
  [rel_pmc_fetch_size.txt]=relation_pmc_fetch_size.txt [rel_pmc_write_size.txt]=relation_pmc_write_size.txt
  [relation_pmc_sq.txt]=relation_pmc_sq.txt [relation_traffic.json]=relation_traffic.json [rel_bench.txt]=relation_kernel_stats.txt
  [window_pmc_sq.txt]=window_pmc_sq.txt [hipblaslt_calibration.txt]=hipblaslt_calibration.txt [train_bench.json]=train_bench.json
  [train_bench_f32.json]=train_bench_f32.json [train_bench_hvr.json]=train_bench_hvr.json [train_kernel_stats.txt]=train_kernel_stats.txt [train_kernel_stats_hvr.txt]=train_kernel_stats_hvr.txt [train_gpu_idle.txt]=train_gpu_idle.txt [train_gpu_idle_hvr.txt]=train_gpu_idle_hvr.txt [train_bench_hvr_inline.json]=train_bench_hvr_inline.json [train_census_hvr.txt]=train_census_hvr.txt [train_census_selsa.txt]=train_census_selsa.txt
  [ingest_bench.json]=ingest_bench.json [bench_selsa.json]=bench_selsa.json [bench_T21.json]=bench_T21.json
  [precision_ladder.json]=precision_ladder.json [window_f16_kernel_stats.txt]=window_f16_kernel_stats.txt
  [window_f16x2_kernel_stats.txt]=window_f16x2_kernel_stats.txt [conv_layer3.txt]=conv_layer3.txt
  [conv_hint_sweep_f16x2.txt]=conv_hint_sweep_f16x2.txt [window_f16x2_pmc_sq.txt]=window_f16x2_pmc_sq.txt [hipblaslt_kernels.txt]=hipblaslt_kernels.txt [stream_bench.json]=stream_bench.json
  [stream_frame_kernel_stats.txt]=stream_frame_kernel_stats.txt [rpn_wide_kernel_stats.txt]=rpn_wide_kernel_stats.txt
  [rel_pmc_fetch_size_g4.txt]=relation_pmc_fetch_size_g4.txt [rel_pmc_write_size_g4.txt]=relation_pmc_write_size_g4.txt
  [relation_pmc_sq_g4.txt]=relation_pmc_sq_g4.txt [relation_traffic_g4.json]=relation_traffic_g4.json [rel_bench_g4.txt]=relation_kernel_stats_g4.txt
  [window_breakdown_bf16.txt]=window_breakdown_bf16.txt [window_breakdown_bf16_w4.txt]=window_breakdown_bf16_w4.txt
  [window_breakdown_f16x2_w4.txt]=window_breakdown_f16x2_w4.txt [rel_bench_f16x2.txt]=relation_kernel_stats_f16x2.txt
  [rel_bench_f16x2_g4.txt]=relation_kernel_stats_f16x2_g4.txt [frame_breakdown.txt]=frame_breakdown.txt [relation_traffic_f16x2_g4.json]=relation_traffic_f16x2_g4.json )
for f in "${!map[@]}"; do
  if [ -s $src/$f ]; then cp $src/$f profiles/${r}_${map[$f]}; fi
done
ls profiles | grep "^${r}_"
