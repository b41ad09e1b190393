#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in "" "--no-prefetch"; do
timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head hvr $a 2>&1 | tail -1 | cut -c1-420
done
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head hvr 2>&1 | tail -1 | cut -c1-300
