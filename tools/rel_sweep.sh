#!/bin/bash
# usage: rel_sweep.sh "S:A" ...   (tile hints for scores:apply)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
for cfg in "$@"; do
  S=${cfg%%:*}; A=${cfg##*:}
  rm -rf /tmp/prof_$S_$A
  HVR_TILE_SCORES=$S HVR_TILE_APPLY=$A rocprofv3 --kernel-trace --stats -d /tmp/prof_${S}_${A} -o r -- python tools/kernel_bench.py --only relation --iters 10 > /tmp/kb_${S}_${A}.txt 2>&1
  DB=$(find /tmp/prof_${S}_${A} -name "*.db" | head -1)
  echo "=== scores=$S apply=$A"; grep "^relation" /tmp/kb_${S}_${A}.txt
  python tools/rocpd_stats.py $DB | grep -v "at::native\|rocclr" | head -12
done
