#!/bin/bash
# Collects the artefacts kept under profiles/ (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of
# the same command, PMC FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only), relation-core passes alone.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
python bench.py --head selsa --steps 20 --warmup 3 --no-train-step > $out/bench_selsa.json 2>/dev/null
python bench.py --frames 21 --steps 10 --warmup 2 --no-train-step --no-f32-leg --no-side-loops --quick > $out/bench_T21.json 2>/dev/null   # the shipped window length (frame_interval = 10)
rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o bench -- python bench.py --steps 5 --warmup 2 --repeats 2 --no-f32-leg --no-cpu-baseline --no-train-step > $out/bench_prof.json 2>/dev/null
python tools/rocpd_stats.py $(find /tmp/p_ks -name "*.db" | head -1) > $out/bench_kernel_stats.txt
python tools/rocpd_phases.py $(find /tmp/p_ks -name "*.db" | head -1) 4 > $out/bench_window_phases.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c; rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o bench -- python bench.py --steps 2 --warmup 1 --repeats 1 --no-f32-leg --no-cpu-baseline --no-train-step > /dev/null 2>&1
  python tools/rocpd_pmc.py $(find /tmp/p_$c -name "*.db" | head -1) $c > $out/pmc_$(echo $c | tr A-Z a-z).txt
  rm -rf /tmp/r_$c; rocprofv3 --kernel-trace --pmc $c -d /tmp/r_$c -o rel -- python tools/rel_bench.py --iters 5 > /dev/null 2>&1
  python tools/rocpd_pmc.py $(find /tmp/r_$c -name "*.db" | head -1) $c > $out/rel_pmc_$(echo $c | tr A-Z a-z).txt
done
python tools/make_traffic_json.py $(find /tmp/r_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/r_WRITE_SIZE -name "*.db" | head -1) $out/relation_traffic.json
# SQ / TCC passes: the relation core alone, then every kernel of the headline window (the conv classes: MFMA-busy, LDS, waits)
rm -f $out/relation_pmc_sq.txt; bash tools/rel_pmc.sh $out/relation_pmc_sq.txt
rm -f $out/window_pmc_sq.txt; i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/w_$i
  HVR_FRAME_GROUPS=1 rocprofv3 --kernel-trace --pmc $set -d /tmp/w_$i -o w -- python bench.py --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-train-step --no-side-loops --no-graphs --no-f32-leg > /dev/null 2>&1
  echo "--- pass $i: $set" >> $out/window_pmc_sq.txt
  python tools/pmc_dump.py $(find /tmp/w_$i -name "*.db" | head -1) _kernel >> $out/window_pmc_sq.txt 2>&1
done
rm -rf /tmp/r_ks; rocprofv3 --kernel-trace --stats -d /tmp/r_ks -o rel -- python tools/rel_bench.py --iters 10 > $out/rel_bench.txt 2>/dev/null
python tools/rocpd_stats.py $(find /tmp/r_ks -name "*.db" | head -1) >> $out/rel_bench.txt
python tools/probe/blaslt_ref.py > $out/hipblaslt_calibration.txt 2>/dev/null
# (which macro tiles the vendor GEMM picks for these shapes: its kernel names carry them)
rm -rf /tmp/bl_ks; rocprofv3 --kernel-trace --stats -d /tmp/bl_ks -o bl -- python tools/probe/blaslt_ref.py > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/bl_ks -name "*.db" | head -1) > $out/hipblaslt_kernels.txt 2>&1
# training step (SELSA, 1 key + 2 ref frames 600x1000, 300 proposals): throughput in both compute modes + kernel stats of the bf16 mode
python tools/train_bench.py --steps 10 --warmup 2 > $out/train_bench.json 2>/dev/null
python tools/train_bench.py --steps 5 --warmup 2 --dtype f32 > $out/train_bench_f32.json 2>/dev/null
python tools/train_bench.py --steps 10 --warmup 2 --head hvr > $out/train_bench_hvr.json 2>/dev/null
rm -rf /tmp/t_ks; rocprofv3 --kernel-trace --stats -d /tmp/t_ks -o train -- python tools/train_bench.py --steps 3 --warmup 1 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/t_ks -name "*.db" | head -1) > $out/train_kernel_stats.txt
python tools/ingest_bench.py > $out/ingest_bench.json 2>/dev/null
# round 3: the precision ladder at full size (every mode against the CPU oracle + window cost), kernel stats of the half / split-half
# windows, and the three convs of a layer-3 Bottleneck alone (timings per format; FETCH / WRITE / SQ passes of the bf16 block)
python tools/precision_ladder.py --modes bf16,f16,f16x2,f32,trunk_f16x2+head_f16 --out $out/precision_ladder.json > /dev/null 2>&1
for m in f16 f16x2; do
  rm -rf /tmp/m_$m; HVR_FRAME_GROUPS=1 rocprofv3 --kernel-trace --stats -d /tmp/m_$m -o w -- python tools/mode_window.py --mode $m --iters 3 > /dev/null 2>&1
  python tools/rocpd_stats.py $(find /tmp/m_$m -name "*.db" | head -1) > $out/window_${m}_kernel_stats.txt
done
rm -f $out/window_f16x2_pmc_sq.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/ms; HVR_FRAME_GROUPS=1 rocprofv3 --kernel-trace --pmc $set -d /tmp/ms -o w -- python tools/mode_window.py --mode f16x2 --iters 1 > /dev/null 2>&1
  echo "--- split-half window, pass: $set" >> $out/window_f16x2_pmc_sq.txt
  python tools/pmc_dump.py $(find /tmp/ms -name "*.db" | head -1) tile_kernel >> $out/window_f16x2_pmc_sq.txt 2>&1
done
rm -f $out/conv_layer3.txt
for d in bf16 f16 f16x2 f32; do python tools/probe/l3_block.py --dtype $d 2>/dev/null | grep "layer-3" >> $out/conv_layer3.txt; done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/l3; rocprofv3 --kernel-trace --pmc $set -d /tmp/l3 -o l3 -- python tools/probe/l3_block.py --dtype bf16 --iters 3 > /dev/null 2>&1
  echo "--- bf16 layer-3 block, pass: $set (FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x 2 on gfx950)" >> $out/conv_layer3.txt
  python tools/pmc_dump.py $(find /tmp/l3 -name "*.db" | head -1) _kernel >> $out/conv_layer3.txt 2>&1
done
python tools/probe/conv_hint_sweep.py --dtype f16x2 > $out/conv_hint_sweep_f16x2.txt 2>/dev/null
