#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/train_bench.py --steps 5 --warmup 3 --head selsa --detail 2>&1 | head -40 | cut -c1-120
