#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for h in hvr selsa; do
echo "== $h new / old, first iteration and third"
for st in 1 3; do
timeout 300 python tools/train_bench.py --steps $st --warmup 0 --head $h 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['last_losses'])"
(cd abtest/old && timeout 300 python tools/train_bench.py --steps $st --warmup 0 --head $h 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['last_losses'])")
done
done
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t21.log 2>&1; tail -5 gpurun_out/t21.log
