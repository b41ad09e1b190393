#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_targets_gpu.py -x -q 2>&1 | tail -15
for h in selsa hvr; do
timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head $h 2>&1 | tail -1 | cut -c1-420
done
