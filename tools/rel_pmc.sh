#!/bin/bash
# PMC passes over the full-stage relation core; usage: rel_pmc.sh <outfile> [env assignments...]   (REL_GROUPS=G: the grouped call, G windows)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1; shift
for kv in "$@"; do export "$kv"; done
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o r -- python tools/rel_bench.py --iters 3 --groups ${REL_GROUPS:-1} > /tmp/pmc_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  echo "--- pass $i: $set" >> $out
  if [ -n "$DB" ]; then python tools/pmc_dump.py $DB ${PMC_FILTER:-_kernel} >> $out 2>&1; else tail -5 /tmp/pmc_$i.log >> $out; fi
done
