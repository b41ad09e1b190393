#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c14; mkdir -p $O
timeout 300 python tools/window_breakdown.py --mode bf16 --clips 4 > $O/window_breakdown_bf16_w4.txt 2>&1
timeout 300 python tools/window_breakdown.py --mode f16x2 --clips 4 > $O/window_breakdown_f16x2_w4.txt 2>&1
timeout 300 python tools/window_breakdown.py --mode bf16 --clips 1 > $O/window_breakdown_bf16.txt 2>&1
head -45 $O/window_breakdown_bf16_w4.txt
