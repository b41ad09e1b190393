#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1; do
  if [ $v = 1 ]; then export HVR_DBG_TAIL_BIG=1; fi
  echo "== HVR_DBG_TAIL_BIG=$v"
  timeout 300 python tools/window_breakdown.py --mode bf16 --clips 4 2>/dev/null | grep -E "per call|tail 38x63"
  timeout 300 python tools/window_breakdown.py --mode bf16 --clips 1 2>/dev/null | grep -E "per call|tail 38x63"
done
