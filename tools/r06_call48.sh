#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head hvr 2>&1 | tail -1 | cut -c150-240
HVR_DBG_ST=1 timeout 300 python tools/train_bench.py --steps 20 --warmup 3 --head hvr 2>&1 | tail -1 | cut -c150-240
done
