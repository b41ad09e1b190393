#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c6; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -3 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
HVR_BIGTILE_WGS=0 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-side-loops --no-graphs > $O/bench_onetile.json 2> $O/bench_onetile.err; echo "bench (one tile per workgroup) rc $?"
