#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_precision_gpu.py -x -q -k "two_level" 2>&1 | tail -12
timeout 600 python tools/window_breakdown.py --mode f16x2 --iters 2 --clips 4 2>/dev/null | grep -E "per call|1024->512 k3|rpn" | head
