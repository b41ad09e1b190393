#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r06_final_$i.log 2>&1; tail -2 gpurun_out/gputest_r06_final_$i.log
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
