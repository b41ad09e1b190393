#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r06_final_4.log 2>&1; tail -3 gpurun_out/gputest_r06_final_4.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ROUND=r06 bash tools/collect_profiles.sh A > /dev/null 2>&1
timeout 1500 python bench.py --steps 20 --warmup 3 --tol-clips 24 --no-train-step --no-side-loops > gpurun_out/bench_tol24_c.json 2> gpurun_out/bench_tol24_c.err
ROUND=r06 bash tools/collect_profiles.sh CD > /dev/null 2>&1
ls gpurun_out/profiles | wc -l
