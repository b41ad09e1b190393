#!/bin/bash
# Collects the artefacts kept under profiles/ (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of
# the same command, PMC FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only), relation-core passes alone.
# Every step runs under its own `timeout` (T): one profiled run that hangs must not eat the GPU budget of the rest (round 3 lost
# 45 GPU-minutes to exactly that).  Parts:  A = bench lines + kernel stats of the bench command + relation-core passes,
# B = window-wide PMC passes, C = training / ingest / vendor calibration, D = precision ladder + per-mode kernel stats + layer 3,
# E = stream mode: pipelined stream bench, per-kernel profile of one frame, the one-frame proposal call (chip-wide kernels), key-stage merge A/B.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh A'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
parts=${1:-ABCDE}
T="timeout 420"
db() { find $1 -name "*.db" | head -1; }

if [[ $parts == *A* ]]; then
# the relation core alone: one window per call (hvr_relation_fwd) and the product's four windows per call (hvr_relation_fwd_grouped)
for G in 1 4; do
  sfx=$([ $G = 1 ] && echo "" || echo "_g$G")
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/r_$c; $T rocprofv3 --kernel-trace --pmc $c -d /tmp/r_$c -o rel -- python tools/rel_bench.py --iters 5 --groups $G > /dev/null 2>&1
    $T python tools/rocpd_pmc.py $(db /tmp/r_$c) $c > $out/rel_pmc_$(echo $c | tr A-Z a-z)$sfx.txt
  done
  $T python tools/make_traffic_json.py $(db /tmp/r_FETCH_SIZE) $(db /tmp/r_WRITE_SIZE) $out/relation_traffic$sfx.json $G
  rm -f $out/relation_pmc_sq$sfx.txt; REL_GROUPS=$G $T bash tools/rel_pmc.sh $out/relation_pmc_sq$sfx.txt
  rm -rf /tmp/r_ks; $T rocprofv3 --kernel-trace --stats -d /tmp/r_ks -o rel -- python tools/rel_bench.py --iters 10 --groups $G > $out/rel_bench$sfx.txt 2>/dev/null
  $T python tools/rocpd_stats.py $(db /tmp/r_ks) >> $out/rel_bench$sfx.txt
done
# the split-half relation core, four windows per call: FETCH / WRITE passes -> the traffic file bench.py reads for the f16x2 mode's roofline
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rs_$c; $T rocprofv3 --kernel-trace --pmc $c -d /tmp/rs_$c -o rel -- python tools/rel_bench.py --iters 5 --groups 4 --dtype f16x2 > /dev/null 2>&1
done
$T python tools/make_traffic_json.py $(db /tmp/rs_FETCH_SIZE) $(db /tmp/rs_WRITE_SIZE) $out/relation_traffic_f16x2_g4.json 4 f16x2
# the split-half relation core (scores on the persistent big tiles since round 5; V^T by its idle workgroups): kernel stats, 1 / 4 windows per call
for G in 1 4; do
  sfx=$([ $G = 1 ] && echo "" || echo "_g$G")
  rm -rf /tmp/r_ks; $T rocprofv3 --kernel-trace --stats -d /tmp/r_ks -o rel -- python tools/rel_bench.py --dtype f16x2 --iters 10 --groups $G > $out/rel_bench_f16x2$sfx.txt 2>/dev/null
  $T python tools/rocpd_stats.py $(db /tmp/r_ks) >> $out/rel_bench_f16x2$sfx.txt
done
# (the traffic files are keyed to this build: put them where bench.py looks before the bench lines are taken)
r=${ROUND:-r06}; cp $out/relation_traffic.json profiles/${r}_relation_traffic.json; cp $out/relation_traffic_g4.json profiles/${r}_relation_traffic_g4.json; cp $out/relation_traffic_f16x2_g4.json profiles/${r}_relation_traffic_f16x2_g4.json
$T python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
$T python bench.py --head selsa --steps 20 --warmup 3 --no-train-step > $out/bench_selsa.json 2>/dev/null
$T python bench.py --frames 21 --steps 10 --warmup 2 --no-train-step --no-f32-leg --no-side-loops --quick > $out/bench_T21.json 2>/dev/null   # the shipped window length (frame_interval = 10)
# kernel stats of the bench command (the eager single-lane region is where bench.py takes the relation core's HIP events; the
# graph legs are skipped under the profiler: a kernel-trace of two graph lanes replaying did not come back in round 3)
rm -rf /tmp/p_ks; $T rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o bench -- python bench.py --steps 8 --warmup 4 --repeats 2 --no-graphs --no-f32-leg --no-cpu-baseline --no-train-step > $out/bench_prof.json 2> $out/bench_prof.err
ROCPD_SPLIT=relation_scores_bt_kernel:100 $T python tools/rocpd_stats.py $(db /tmp/p_ks) > $out/bench_kernel_stats.txt
$T python tools/rocpd_phases.py $(db /tmp/p_ks) 4 > $out/bench_window_phases.txt 2>&1
$T python tools/window_breakdown.py --mode bf16 --iters 3 > $out/window_breakdown_bf16.txt 2>/dev/null
$T python tools/window_breakdown.py --mode bf16 --iters 3 --clips 4 > $out/window_breakdown_bf16_w4.txt 2>/dev/null
$T python tools/window_breakdown.py --mode f16x2 --iters 2 --clips 4 > $out/window_breakdown_f16x2_w4.txt 2>/dev/null
fi

if [[ $parts == *B* ]]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c; $T rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o bench -- python bench.py --steps 2 --warmup 1 --repeats 1 --no-graphs --no-side-loops --no-f32-leg --no-cpu-baseline --no-train-step > /dev/null 2>&1
  $T python tools/rocpd_pmc.py $(db /tmp/p_$c) $c > $out/pmc_$(echo $c | tr A-Z a-z).txt
done
# SQ / TCC passes over every kernel of the headline window (the conv classes: MFMA-busy, LDS, waits)
rm -f $out/window_pmc_sq.txt; i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/w_$i
  $T rocprofv3 --kernel-trace --pmc $set -d /tmp/w_$i -o w -- python bench.py --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-train-step --no-side-loops --no-graphs --no-f32-leg > /dev/null 2>&1
  echo "--- pass $i: $set" >> $out/window_pmc_sq.txt
  $T python tools/pmc_dump.py $(db /tmp/w_$i) _kernel >> $out/window_pmc_sq.txt 2>&1
done
fi

if [[ $parts == *C* ]]; then
$T python tools/blaslt_ref.py > $out/hipblaslt_calibration.txt 2>/dev/null
# (which macro tiles the vendor GEMM picks for these shapes: its kernel names carry them)
rm -rf /tmp/bl_ks; $T rocprofv3 --kernel-trace --stats -d /tmp/bl_ks -o bl -- python tools/blaslt_ref.py > /dev/null 2>&1
$T python tools/rocpd_stats.py $(db /tmp/bl_ks) > $out/hipblaslt_kernels.txt 2>&1
# training step (SELSA, 1 key + 2 ref frames 600x1000, 300 proposals): throughput in both compute modes + kernel stats of the bf16 mode
$T python tools/train_bench.py --steps 20 --warmup 3 > $out/train_bench.json 2>/dev/null
$T python tools/train_bench.py --steps 5 --warmup 2 --dtype f32 > $out/train_bench_f32.json 2>/dev/null
$T python tools/train_bench.py --steps 20 --warmup 3 --head hvr > $out/train_bench_hvr.json 2>/dev/null                       # the frozen backbone one batch ahead (dist_train.C4Prefetcher)
$T python tools/train_bench.py --steps 20 --warmup 3 --head hvr --no-prefetch > $out/train_bench_hvr_inline.json 2>/dev/null   # ... and in line
# kernel stats over 40 iterations per head (the model build and the first, recording iteration are ~2 % of the dispatches), and the GPU's
# busy / idle time per steady-state iteration (tools/gpu_idle.py; stem_fused_kernel marks an iteration)
for h in selsa hvr; do
  sfx=$([ $h = selsa ] && echo "" || echo "_hvr")
  rm -rf /tmp/t_ks; $T rocprofv3 --kernel-trace --stats -d /tmp/t_ks -o train -- python tools/train_bench.py --steps 38 --warmup 2 --head $h > /dev/null 2>&1
  $T python tools/rocpd_stats.py $(db /tmp/t_ks) > $out/train_kernel_stats$sfx.txt
  $T python tools/gpu_idle.py $(db /tmp/t_ks) stem_fused_kernel 10 20 > $out/train_gpu_idle$sfx.txt 2>&1
done
$T python tools/train_census.py --head hvr > $out/train_census_hvr.txt 2>/dev/null
$T python tools/train_census.py --head selsa > $out/train_census_selsa.txt 2>/dev/null
$T python tools/ingest_bench.py > $out/ingest_bench.json 2>/dev/null
fi

if [[ $parts == *D* ]]; then
# round 3: the precision ladder at full size (every mode against the CPU oracle + window cost), kernel stats of the half / split-half
# windows, and the three convs of a layer-3 Bottleneck alone (timings per format; FETCH / WRITE / SQ passes of the bf16 block)
timeout 900 python tools/precision_ladder.py --modes bf16,f16,f16x2,f32,trunk_f16x2+head_f16 --out $out/precision_ladder.json > /dev/null 2>&1
for m in f16 f16x2; do
  rm -rf /tmp/m_$m; $T rocprofv3 --kernel-trace --stats -d /tmp/m_$m -o w -- python tools/mode_window.py --mode $m --iters 3 > /dev/null 2>&1
  $T python tools/rocpd_stats.py $(db /tmp/m_$m) > $out/window_${m}_kernel_stats.txt
done
rm -f $out/window_f16x2_pmc_sq.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/ms; $T rocprofv3 --kernel-trace --pmc $set -d /tmp/ms -o w -- python tools/mode_window.py --mode f16x2 --iters 1 > /dev/null 2>&1
  echo "--- split-half window, pass: $set" >> $out/window_f16x2_pmc_sq.txt
  $T python tools/pmc_dump.py $(db /tmp/ms) tile_kernel >> $out/window_f16x2_pmc_sq.txt 2>&1
done
rm -f $out/conv_layer3.txt
for d in bf16 f16 f16x2 f32; do $T python tools/l3_block.py --dtype $d 2>/dev/null | grep "layer-3" >> $out/conv_layer3.txt; done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  rm -rf /tmp/l3; $T rocprofv3 --kernel-trace --pmc $set -d /tmp/l3 -o l3 -- python tools/l3_block.py --dtype bf16 --iters 3 > /dev/null 2>&1
  echo "--- bf16 layer-3 block, pass: $set (FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x 2 on gfx950)" >> $out/conv_layer3.txt
  $T python tools/pmc_dump.py $(db /tmp/l3) _kernel >> $out/conv_layer3.txt 2>&1
done
$T python tools/conv_hint_sweep.py --dtype f16x2 > $out/conv_hint_sweep_f16x2.txt 2>/dev/null
fi
ls $out

if [[ $parts == *E* ]]; then
$T python tools/stream_bench.py --steps 60 2>/dev/null | tail -1 > $out/stream_bench.json
$T python tools/frame_breakdown.py 2>/dev/null > $out/frame_breakdown.txt
{ echo; echo "# the same with HVR_CONV_SPLITK=2 (K sliced across workgroups + a reduce launch everywhere, rounds 3-4's form):"; HVR_CONV_SPLITK=2 $T python tools/frame_breakdown.py 2>/dev/null; } >> $out/frame_breakdown.txt
rm -rf /tmp/f_ks; $T rocprofv3 --kernel-trace --stats -d /tmp/f_ks -o fr -- python tools/frame_profile.py 20 > $out/stream_frame.log 2>&1
{ echo "# tools/frame_profile.py under rocprofv3 --kernel-trace --stats: 23 frames (3 warm-up + 20), eager; per-frame rows = call counts that are multiples of 23 (the rest is one-time weight packing)"; grep "one frame" $out/stream_frame.log; $T python tools/rocpd_stats.py $(db /tmp/f_ks) | head -45; } > $out/stream_frame_kernel_stats.txt
rm -rf /tmp/rw_ks; HVR_RPN_WIDE=4 $T rocprofv3 --kernel-trace --stats -d /tmp/rw_ks -o rpn -- python tools/rpn_probe.py > $out/rpn_wide_probe.txt 2>&1
$T python tools/rocpd_stats.py $(db /tmp/rw_ks) | grep -i "rpn\|nms\|kernel \|dispatches" > $out/rpn_wide_kernel_stats.txt
$T bash tools/run_stream_ab.sh > /dev/null 2>&1   # -> gpurun_out/stream_ab.txt
fi
