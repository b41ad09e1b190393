#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in "" "--overlap"; do
timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head selsa $a 2>&1 | tail -1 | cut -c1-300
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head selsa $a 2>&1 | tail -1 | cut -c1-300
done
timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head hvr --overlap 2>&1 | tail -1 | cut -c1-300
