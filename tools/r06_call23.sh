#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_targets_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -3
for h in hvr selsa; do
for cfg in "0 160" "1 160" "1 256" "1 400"; do
set -- $cfg
echo "== $h fewrow=$1 tiles<=$2"
HVR_DBG_TRAIN_FEWROW=$1 HVR_DBG_FEWROW_TILES=$2 timeout 300 python tools/train_bench.py --steps 10 --warmup 2 --head $h 2>&1 | tail -1 | cut -c1-330
done
done
