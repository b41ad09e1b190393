#!/bin/bash
# Round 6, GPU call 1: baseline tests + bench with the eight-clip tolerance claim, the expand-kernel ablation table, the big-tile
# per-round probe + clock stamps, PMC passes over layer 3's expand / reducing convs.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c1; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ) 
tail -5 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
# ---- expand kernel (row-panel, expand.hip) ablations: throughput hint rows = expand.hip at l3
echo "== product build" > $O/expand_ablation.txt
timeout 300 python tools/expand_bench.py --stages l3 --modes bf16 --frames 15,60 >> $O/expand_ablation.txt 2>&1
for v in NOMMA NODMA NORES NOSTORE NOLDS; do
  echo "== -DHVR_DBG_X_$v (timing only)" >> $O/expand_ablation.txt
  HVR_BENCH_LIB=abtest/libhvr_x_$v.so timeout 300 python tools/expand_bench.py --stages l3 --modes bf16 --frames 15,60 >> $O/expand_ablation.txt 2>&1
done
# ---- big tiles: per-round cost and per-phase clock stamps
timeout 600 python tools/bigtile_probe.py --frames 15,26,30,52,60,104,120 --shapes reduce,c3,expand,res5c3,rpn > $O/bigtile_probe.txt 2>&1
HVR_BENCH_LIB=abtest/libhvr_bgclk.so timeout 300 python tools/bigtile_probe.py --clk --frames 30,60 --shapes reduce,c3,expand > $O/bigtile_clk.txt 2>&1
# ---- PMC passes: layer 3 expand (expand.hip under the throughput hint, big tiles by default) and reducing 1x1 (big tiles), 60 frames
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o r -- python tools/expand_bench.py --stages l3 --modes bf16 --frames 60 --iters 3 > /tmp/pmc_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "--- pass $i (expand_bench l3 bf16 60 frames): $set" >> $O/conv_pmc.txt
  if [ -n "$DB" ]; then python tools/pmc_dump.py $DB _kernel >> $O/conv_pmc.txt 2>&1; else tail -5 /tmp/pmc_$i.log >> $O/conv_pmc.txt; fi
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o r -- python tools/bigtile_probe.py --frames 60 --shapes reduce,c3 --iters 3 > /tmp/pmc_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "--- pass $i (bigtile_probe reduce,c3 60 frames): $set" >> $O/conv_pmc.txt
  if [ -n "$DB" ]; then python tools/pmc_dump.py $DB _kernel >> $O/conv_pmc.txt 2>&1; else tail -5 /tmp/pmc_$i.log >> $O/conv_pmc.txt; fi
done
echo done
